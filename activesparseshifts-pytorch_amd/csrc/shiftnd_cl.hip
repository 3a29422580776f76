// shiftnd_cl.hip -- channel-fastest kernels for channels-last (NHWC / NDHWC) tensors (gfx950).
//
// The reference's CUDA backend has no channels-last kernel (its NCHW kernel is run with the tensor's
// strides, cuda/shifts_cuda.cu:217-262); its CPU backend has one (kernels/shifts_kernels.h:330-527:
// inner loop over C).  Here a channels-last input is served by kernels that iterate the way the data
// lies: the lanes of a wave are consecutive CHANNELS of one pixel, so the stores (channels-last output /
// grad_x) are whole contiguous segments and the loads are gathers inside the few neighbouring pixels the
// channels' shifts point to (they share cache lines across lanes with equal shifts and across the
// neighbouring pixels' waves through L2).
//
//   * A workgroup owns CW consecutive channels x PL pixel lanes; a thread keeps ONE channel for its
//     whole life, so the per-channel work (weight -> integer shift + fraction, canonical shift of the
//     padding map) is done once in the prologue.
//   * The padding map is the arithmetic one of the sweep kernels (canon_shift + fold_index, 32-bit,
//     no tables, no division): per element a handful of compares.
//   * Tensors are addressed through their element strides, so the float forward's NCHW-contiguous
//     output (cpu/shifts_cpu.cpp:221: the reference allocates it contiguous even for a channels-last
//     input), arbitrary grad_out layouts and channels-last grad_x all work; only the iteration order is
//     specialised.
//   * grad_w: per-thread fp64 accumulation over the thread's pixels, summed over the PL pixel lanes in
//     LDS (fixed order), one partial per (pixel group, channel); reduce_weight_grads finishes: deterministic.
//
// Reference behaviour restated: forward kernels/shifts_kernels.h:330-400 (nhwdc) == :156-220 values;
// backward :402-527 == :222-327; quantized :574-624 == :532-571.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

struct ClParams {
    const void *x, *go, *w;
    void *out;           // forward: output; backward: grad_x
    double *partials;    // backward: [pgroups][C][3]
    int64_t wzp;
    uint64_t fill;
    int64_t xs_n, os_n, gs_n;  // batch strides (64-bit); everything inside one sample is addressed with 32-bit offsets
    int xs[4], os[4], gs[4];   // element strides C, d0, d1, inner of x / out or grad_out / grad_x
    int wkind, N, C, nd, pad;
    int S[3], O[3], L[3], wcol[3];
    int CW, PL, cgroups, pgroups;
    int seg, nseg;       // a work item = `seg` consecutive inner positions of one row (n, i0, i1)
    uint32_t items;      // rows * nseg
    FastDiv d_nseg, d_rows, d_i1;  // item -> row; row -> n (rows per sample); -> i0 (extent of i1)
    FastDiv d_per[3];    // period dividers of the x maps (sizes S)
    FastDiv d_gper[3];   // period dividers of the grad_out maps (sizes O)
};

struct Lane {
    int c;        // this thread's channel
    int tp;       // pixel lane
    bool live;
};
__device__ __forceinline__ Lane decode_lane(const ClParams &p, int &pgrp) {
    Lane l;
    const int cgrp = static_cast<int>(blockIdx.x) % p.cgroups;
    pgrp = static_cast<int>(blockIdx.x) / p.cgroups;
    l.tp = static_cast<int>(threadIdx.x) / p.CW;
    l.c = cgrp * p.CW + static_cast<int>(threadIdx.x) - l.tp * p.CW;
    l.live = l.c < p.C && l.tp < p.PL;
    return l;
}

struct Item {  // one row segment of the iteration space with extents ext[3]
    int n, i0, i1, j0, j1;  // inner positions [j0, j1)
};
__device__ __forceinline__ Item decode_item(const ClParams &p, uint32_t item, const int ext[3]) {
    Item t;
    const uint32_t row = fdiv(item, p.d_nseg);
    const int sg = static_cast<int>(item - row * static_cast<uint32_t>(p.nseg));
    t.n = static_cast<int>(fdiv(row, p.d_rows));
    const int r = static_cast<int>(row - static_cast<uint32_t>(t.n) * static_cast<uint32_t>(ext[0] * ext[1]));
    t.i0 = static_cast<int>(fdiv(static_cast<uint32_t>(r), p.d_i1));
    t.i1 = r - t.i0 * ext[1];
    t.j0 = sg * p.seg;
    t.j1 = min(ext[2], t.j0 + p.seg);
    return t;
}

// padded-map source coordinate along one dim (size-1 dims ignore the shift, shifts_kernels.h:40)
__device__ __forceinline__ int fold_dim(int idx, int size, int pad) { return size == 1 ? 0 : fold_index(idx, size, pad); }

// row offsets (elements within the sample and channel) of the outer-dim corner combos of a row:
// combo k bit r <-> +1 along real dim r (r < ND-1); -1 = fill.  For ND == 1 there is one combo (no outer real dim).
template <int ND>
__device__ __forceinline__ void corner_rows(int a, int b, const int size[3], const int st[4], int pad, int (&off)[1 << (ND - 1)]) {
#pragma unroll
    for (int k = 0; k < (1 << (ND - 1)); ++k) {
        const int ha = ND == 3 ? (k & 1) : 0;
        const int hb = ND == 3 ? ((k >> 1) & 1) : (ND == 2 ? (k & 1) : 0);
        const int ra = fold_dim(a + ha, size[0], pad), rb = fold_dim(b + hb, size[1], pad);
        off[k] = (ra < 0 || rb < 0) ? -1 : ra * st[1] + rb * st[2];
    }
}

// ---- SSL / quantized forward: pure gather -------------------------------------------------------------------
template <int ESIZE>
__global__ __launch_bounds__(kThreads) void cl_gather_forward(const ClParams p) {
    using R = typename raw_t<ESIZE>::type;
    int pgrp;
    const Lane l = decode_lane(p, pgrp);
    if (!l.live) return;
    int cs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        cs[d] = p.wcol[d] >= 0 ? canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(l.c) * p.nd + p.wcol[d]), p.S[d], p.pad, p.d_per[d]) : 0;
    const R *x = static_cast<const R *>(p.x) + l.c * p.xs[0];
    R *out = static_cast<R *>(p.out) + l.c * p.os[0];
    const R fill = static_cast<R>(p.fill);
    const uint32_t step = static_cast<uint32_t>(p.pgroups) * p.PL;
    for (uint32_t item = static_cast<uint32_t>(pgrp) * p.PL + l.tp; item < p.items; item += step) {
        const Item t = decode_item(p, item, p.O);
        const int ra = fold_dim(t.i0 + p.L[0] - cs[0], p.S[0], p.pad), rb = fold_dim(t.i1 + p.L[1] - cs[1], p.S[1], p.pad);
        const bool rowok = ra >= 0 && rb >= 0;
        const R *xrow = x + t.n * p.xs_n + (rowok ? ra * p.xs[1] + rb * p.xs[2] : 0);
        R *orow = out + t.n * p.os_n + t.i0 * p.os[1] + t.i1 * p.os[2];
#pragma unroll 4
        for (int j = t.j0; j < t.j1; ++j) {
            const int rc = fold_dim(j + p.L[2] - cs[2], p.S[2], p.pad);
            const R v = xrow[rc >= 0 ? rc * p.xs[3] : 0];
            orow[j * p.os[3]] = (rowok && rc >= 0) ? v : fill;
        }
    }
}

// ---- active forward ----------------------------------------------------------------------------------------
template <typename T, int ND>
__global__ __launch_bounds__(kThreads) void cl_active_forward(const ClParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int NC = 1 << (ND - 1);
    int pgrp;
    const Lane l = decode_lane(p, pgrp);
    if (!l.live) return;
    int cs[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0) {
            int64_t sh;
            prep_shift_forward<CT>(load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(l.c) * p.nd + p.wcol[d]), true, sh, dw[p.wcol[d]]);
            cs[d] = canon_shift(sh, p.S[d], p.pad, p.d_per[d]);
        }
    const S *x = static_cast<const S *>(p.x) + l.c * p.xs[0];
    S *out = static_cast<S *>(p.out) + l.c * p.os[0];
    const uint32_t step = static_cast<uint32_t>(p.pgroups) * p.PL;
    for (uint32_t item = static_cast<uint32_t>(pgrp) * p.PL + l.tp; item < p.items; item += step) {
        const Item t = decode_item(p, item, p.O);
        int roff[NC];
        corner_rows<ND>(t.i0 + p.L[0] - cs[0], t.i1 + p.L[1] - cs[1], p.S, p.xs, p.pad, roff);
        const S *xn = x + t.n * p.xs_n;
        S *orow = out + t.n * p.os_n + t.i0 * p.os[1] + t.i1 * p.os[2];
        for (int j = t.j0; j < t.j1; ++j) {
            const int c0 = fold_dim(j + p.L[2] - cs[2], p.S[2], p.pad), c1 = fold_dim(j + p.L[2] - cs[2] + 1, p.S[2], p.pad);
            CT v[1 << ND];
#pragma unroll
            for (int q = 0; q < (1 << ND); ++q) {
                const int ro = roff[q & (NC - 1)], cc = (q >> (ND - 1)) ? c1 : c0;
                const S raw = xn[(ro >= 0 ? ro : 0) + (cc >= 0 ? cc * p.xs[3] : 0)];
                v[q] = (ro >= 0 && cc >= 0) ? widen<T>(raw) : CT(0);
            }
            orow[j * p.os[3]] = narrow<T>(interp_t<T, ND>(v, dw));
        }
    }
}

// ---- backward ----------------------------------------------------------------------------------------------
template <typename T, int ND, bool ACTIVE>
__global__ __launch_bounds__(kThreads) void cl_backward(const ClParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int NC = 1 << (ND - 1);
    __shared__ double red[kThreads][3];
    int pgrp;
    const Lane l = decode_lane(p, pgrp);
    double acc[3] = {0.0, 0.0, 0.0};
    if (l.live) {
        int csx[3] = {0, 0, 0}, csg[3] = {0, 0, 0};
        CT dw[3] = {CT(0), CT(0), CT(0)};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (p.wcol[d] >= 0) {
                int64_t sh;
                prep_shift_backward<CT>(load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(l.c) * p.nd + p.wcol[d]), ACTIVE, sh, dw[p.wcol[d]]);
                csx[d] = canon_shift(sh, p.S[d], p.pad, p.d_per[d]);
                // grad_x source: SSL reads grad_out at o + shift, active at o - shift (shifts_kernels.h:287-293),
                // padded over the CROPPED sizes (:295-297)
                csg[d] = canon_shift(ACTIVE ? sh : -sh, p.O[d], p.pad, p.d_gper[d]);
            }
        const S *x = static_cast<const S *>(p.x) + l.c * p.xs[0];
        const S *go = static_cast<const S *>(p.go) + l.c * p.os[0];
        S *gx = static_cast<S *>(p.out) + l.c * p.gs[0];
        const uint32_t step = static_cast<uint32_t>(p.pgroups) * p.PL;
        for (uint32_t item = static_cast<uint32_t>(pgrp) * p.PL + l.tp; item < p.items; item += step) {
            const Item t = decode_item(p, item, p.S);
            const int o0 = t.i0 - p.L[0], o1 = t.i1 - p.L[1];
            const bool rowin = o0 >= 0 && o0 < p.O[0] && o1 >= 0 && o1 < p.O[1];
            const S *xn = x + t.n * p.xs_n;
            const S *gon = go + t.n * p.os_n;
            S *gxrow = gx + t.n * p.gs_n + t.i0 * p.gs[1] + t.i1 * p.gs[2];
            int xoff[NC], goff[ACTIVE ? NC : 1];
            corner_rows<ND>(t.i0 - csx[0], t.i1 - csx[1], p.S, p.xs, p.pad, xoff);
            if constexpr (ACTIVE) {
                corner_rows<ND>(o0 - csg[0], o1 - csg[1], p.O, p.os, p.pad, goff);
            } else {
                const int ra = fold_dim(o0 - csg[0], p.O[0], p.pad), rb = fold_dim(o1 - csg[1], p.O[1], p.pad);
                goff[0] = (ra < 0 || rb < 0) ? -1 : ra * p.os[1] + rb * p.os[2];
            }
            const int gdirect = rowin ? o0 * p.os[1] + o1 * p.os[2] : 0;
            for (int j = t.j0; j < t.j1; ++j) {
                const int o2 = j - p.L[2];
                const bool pass = rowin && o2 >= 0 && o2 < p.O[2];
                const S graw = gon[gdirect + (pass ? o2 * p.os[3] : 0)];
                const CT gval = pass ? widen<T>(graw) : CT(0);
                const int xc0 = fold_dim(j - csx[2], p.S[2], p.pad), xc1 = fold_dim(j - csx[2] + 1, p.S[2], p.pad);
                CT v[1 << ND], wg[3];
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) {
                    const int ro = xoff[q & (NC - 1)], cc = (q >> (ND - 1)) ? xc1 : xc0;
                    const S raw = xn[(ro >= 0 ? ro : 0) + (cc >= 0 ? cc * p.xs[3] : 0)];
                    v[q] = (pass && ro >= 0 && cc >= 0) ? widen<T>(raw) : CT(0);
                }
                weight_grads_nd<ND, CT>(v, dw, wg);
                if (pass) {
#pragma unroll
                    for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval * wg[s]);
                }
                if constexpr (ACTIVE) {
                    const int gc0 = fold_dim(o2 - csg[2], p.O[2], p.pad), gc1 = fold_dim(o2 - csg[2] + 1, p.O[2], p.pad);
#pragma unroll
                    for (int q = 0; q < (1 << ND); ++q) {
                        const int ro = goff[q & (NC - 1)], cc = (q >> (ND - 1)) ? gc1 : gc0;
                        const bool ok = pass && ro >= 0 && cc >= 0;
                        const S raw = gon[ok ? ro + cc * p.os[3] : 0];
                        v[q] = ok ? widen<T>(raw) : CT(0);
                    }
                    gxrow[j * p.gs[3]] = narrow<T>(pass ? interp_t<T, ND>(v, dw) : CT(0));
                } else {
                    const int gc = pass ? fold_dim(o2 - csg[2], p.O[2], p.pad) : -1;
                    const bool ok = pass && goff[0] >= 0 && gc >= 0;
                    const S raw = gon[ok ? goff[0] + gc * p.os[3] : 0];
                    gxrow[j * p.gs[3]] = ok ? raw : narrow<T>(CT(0));  // pure copy: keep the bit pattern
                }
            }
        }
    }
    // sum over the pixel lanes of each channel lane, in lane order
#pragma unroll
    for (int s = 0; s < 3; ++s) red[threadIdx.x][s] = acc[s];
    __syncthreads();
    if (l.live && l.tp == 0) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double t = 0.0;
            for (int k = 0; k < p.PL; ++k) t += red[k * p.CW + threadIdx.x][s];
            p.partials[(static_cast<size_t>(pgrp) * p.C + l.c) * 3 + s] = t;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
struct ClPlan {
    int CW, PL, cgroups, pgroups, seg, nseg;
    int64_t items;
};
constexpr int kSegment = 32;  // inner positions per work item
ClPlan cl_plan(const Geometry &g, const int64_t ext[3], int target_wgs) {
    ClPlan pl;
    pl.CW = static_cast<int>(g.C < kThreads ? g.C : kThreads);
    pl.PL = kThreads / pl.CW;
    pl.cgroups = static_cast<int>((g.C + pl.CW - 1) / pl.CW);
    pl.seg = static_cast<int>(ext[2] < kSegment ? ext[2] : kSegment);
    if (pl.seg < 1) pl.seg = 1;
    pl.nseg = static_cast<int>((ext[2] + pl.seg - 1) / pl.seg);
    pl.items = g.N * ext[0] * ext[1] * pl.nseg;
    int64_t pg = (pl.items + pl.PL - 1) / pl.PL;
    const int64_t cap = target_wgs / pl.cgroups > 0 ? target_wgs / pl.cgroups : 1;
    if (pg > cap) pg = cap;
    if (pg < 1) pg = 1;
    pl.pgroups = static_cast<int>(pg);
    return pl;
}

// 32-bit addressing inside one sample: every (size - 1) * stride sum of a tensor stays below 2^31
bool sample_fits(const int64_t st[5], int64_t C, const int64_t sz[3]) {
    int64_t ext = (C - 1) * (st[1] < 0 ? -st[1] : st[1]);
    for (int d = 0; d < 3; ++d) {
        if (st[2 + d] < 0) return false;
        ext += (sz[d] - 1) * st[2 + d];
    }
    return st[1] >= 0 && ext < (1LL << 31) - 64;
}

bool cl_sizes_ok(const Geometry &g, const int64_t ext[3]) {
    if (g.C < 1 || g.C >= (1LL << 30) || g.N >= (1LL << 31)) return false;
    for (int d = 0; d < 3; ++d)
        if (g.S[d] >= (1LL << 30) || g.O[d] >= (1LL << 30)) return false;
    const int64_t items = g.N * ext[0] * ext[1] * ((ext[2] + kSegment - 1) / kSegment + 1);
    return items >= 1 && items < (1LL << 31) && ext[0] * ext[1] < (1LL << 31);
}

// channel-fastest iteration pays when the channel stride of x is the smallest one
bool channel_fastest(const int64_t st[5], const int64_t sz[3], int64_t C) {
    if (C < 2 || st[1] != 1) return false;
    for (int d = 0; d < 3; ++d)
        if (sz[d] > 1 && st[2 + d] < C) return false;
    return true;
}

void fill_cl(ClParams &p, const Geometry &g, const ClPlan &pl, const int64_t ext[3]) {
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.O[d] = static_cast<int>(g.O[d]);
        p.L[d] = static_cast<int>(g.L[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
        p.d_gper[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.O[d], g.pad)));
    }
    p.xs_n = g.xs[0];
    p.os_n = g.os[0];
    p.gs_n = g.gs[0];
    for (int i = 0; i < 4; ++i) {
        p.xs[i] = static_cast<int>(g.xs[1 + i]);
        p.os[i] = static_cast<int>(g.os[1 + i]);
        p.gs[i] = static_cast<int>(g.gs[1 + i]);
    }
    p.CW = pl.CW;
    p.PL = pl.PL;
    p.cgroups = pl.cgroups;
    p.pgroups = pl.pgroups;
    p.seg = pl.seg;
    p.nseg = pl.nseg;
    p.items = static_cast<uint32_t>(pl.items);
    p.d_nseg = make_fastdiv(static_cast<uint32_t>(pl.nseg));
    p.d_rows = make_fastdiv(static_cast<uint32_t>(ext[0] * ext[1]));
    p.d_i1 = make_fastdiv(static_cast<uint32_t>(ext[1]));
}

constexpr int kForwardWgs = 8192, kBackwardWgs = 2048;

}  // namespace

// automatic choice (policy 0): every tensor of the call is channel-fastest.  With mixed layouts (the float forward's
// NCHW-contiguous output for a channels-last input, an NCHW grad_out) one side is uncoalesced whichever way the
// kernel iterates, and the pixel-fastest strided kernels measured faster (N16 C256 224x224 fp32: 3.4 vs 5.9 ms forward).
bool cl_forward_preferred(const Geometry &g) { return cl_forward_eligible(g) && channel_fastest(g.os, g.O, g.C); }
bool cl_backward_preferred(const Geometry &g, int dtype) {
    return cl_backward_eligible(g, dtype) && channel_fastest(g.os, g.O, g.C) && channel_fastest(g.gs, g.S, g.C);
}

bool cl_forward_eligible(const Geometry &g) {
    return channel_fastest(g.xs, g.S, g.C) && cl_sizes_ok(g, g.O) && sample_fits(g.xs, g.C, g.S) && sample_fits(g.os, g.C, g.O);
}

int cl_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
               void *out, hipStream_t st) {
    const ClPlan pl = cl_plan(g, g.O, kForwardWgs);
    ClParams p{};
    p.x = x;
    p.w = w;
    p.out = out;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = fill_bits;
    fill_cl(p, g, pl, g.O);
    const dim3 grid(static_cast<unsigned>(pl.cgroups) * pl.pgroups), block(kThreads);
    const bool gather_only = !g.active || dtype >= SHIFTND_I8;
    note_kernel(gather_only ? "cl_gather_forward" : "cl_active_forward");
    if (gather_only) {
        switch (dtype_size(dtype)) {
        case 1: hipLaunchKernelGGL((cl_gather_forward<1>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((cl_gather_forward<2>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((cl_gather_forward<4>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((cl_gather_forward<8>), grid, block, 0, st, p); break;
        }
        return SHIFTND_OK;
    }
#define SHIFTND_CL_ACTIVE(TT) \
    switch (g.nd) { \
    case 1: hipLaunchKernelGGL((cl_active_forward<TT, 1>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((cl_active_forward<TT, 2>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((cl_active_forward<TT, 3>), grid, block, 0, st, p); break; \
    }
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_CL_ACTIVE(f32_t) break;
    case SHIFTND_F64: SHIFTND_CL_ACTIVE(f64_t) break;
    case SHIFTND_F16: SHIFTND_CL_ACTIVE(f16_t) break;
    case SHIFTND_BF16: SHIFTND_CL_ACTIVE(bf16_t) break;
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
#undef SHIFTND_CL_ACTIVE
    return SHIFTND_OK;
}

bool cl_backward_eligible(const Geometry &g, int dtype) {
    return dtype <= SHIFTND_BF16 && channel_fastest(g.xs, g.S, g.C) && cl_sizes_ok(g, g.S) && sample_fits(g.xs, g.C, g.S) &&
           sample_fits(g.os, g.C, g.O) && sample_fits(g.gs, g.C, g.S);
}

size_t cl_backward_workspace(const Geometry &g) {
    if (g.C < 1 || g.N * g.S[0] * g.S[1] * g.S[2] < 1) return 0;
    const ClPlan pl = cl_plan(g, g.S, kBackwardWgs);
    return static_cast<size_t>(pl.pgroups) * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

template <typename T>
static int launch_cl_backward(const ClParams &p, const Geometry &g, const ClPlan &pl, void *gw, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(pl.cgroups) * pl.pgroups), block(kThreads);
#define SHIFTND_CL_BWD(NDV) \
    if (g.active) hipLaunchKernelGGL((cl_backward<T, NDV, true>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((cl_backward<T, NDV, false>), grid, block, 0, st, p);
    switch (g.nd) {
    case 1: SHIFTND_CL_BWD(1) break;
    case 2: SHIFTND_CL_BWD(2) break;
    default: SHIFTND_CL_BWD(3) break;
    }
#undef SHIFTND_CL_BWD
    reduce_weight_grads_of<T>(p.partials, pl.pgroups, p.C, p.nd, gw, st);
    return SHIFTND_OK;
}

int cl_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                void *workspace, hipStream_t st) {
    const ClPlan pl = cl_plan(g, g.S, kBackwardWgs);
    ClParams p{};
    p.x = x;
    p.go = go;
    p.w = w;
    p.out = gx;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_cl(p, g, pl, g.S);
    note_kernel("cl_backward");
    switch (dtype) {
    case SHIFTND_F32: return launch_cl_backward<f32_t>(p, g, pl, gw, st);
    case SHIFTND_F64: return launch_cl_backward<f64_t>(p, g, pl, gw, st);
    case SHIFTND_F16: return launch_cl_backward<f16_t>(p, g, pl, gw, st);
    case SHIFTND_BF16: return launch_cl_backward<bf16_t>(p, g, pl, gw, st);
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
}

}  // namespace shiftnd
