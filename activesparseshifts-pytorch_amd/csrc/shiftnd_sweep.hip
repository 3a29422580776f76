// shiftnd_sweep.hip -- "sweep" kernels: the HBM-rate path of the shiftnd op on gfx950 (MI355X).
//
// Measured on MI355X (tools/hbm_bench.hip): a wave that performs ONE round of loads followed by ONE
// 16-byte store and then retires streams at 6.4-6.6 TB/s, while workgroups that loop over a private
// region (one plane each) saturate at 5.2-5.5 TB/s however the loop is pipelined.  The sweep kernels
// therefore give every thread exactly one 16-byte output chunk:
//   * the grid is the list of output chunks in memory order, 256 per workgroup, no loops, no LDS maps,
//     no barriers in the forward kernels;
//   * blockIdx is remapped so that the workgroups of one XCD (blockIdx % 8) cover a contiguous eighth
//     of the tensor: rows re-read by neighbouring workgroups (interpolation corners, the shifted
//     grad_out row) hit that XCD's L2 instead of travelling twice from HBM;
//   * loads and stores are nontemporal (every byte is touched once);
//   * the padding map is evaluated arithmetically per element: the channel's shift is reduced to a
//     canonical representative once per thread (canon_shift), after which any index needs at most
//     two conditional folds (fold_index) -- exact for all five modes and any shift magnitude;
//   * a chunk whose E (+1) source columns are consecutive takes one element-aligned 16-byte load,
//     otherwise (chunks touching the padded region) E masked element loads.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/):
//   forward   kernels/shifts_kernels.h:156-220, cuda/shifts_cuda.cu:202-266
//   backward  kernels/shifts_kernels.h:222-327, cuda/shifts_cuda.cu:270-345
//   quantized kernels/shifts_kernels.h:532-571, quantized/shifts_quantized.cpp:107-130
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

#ifndef SWEEP_NT_LOAD
#define SWEEP_NT_LOAD true
#endif
constexpr int kSweepMaxThreads = 512;
thread_local int g_sweep_tune[4] = {4, 512, 2, 256};  // [0] forward row steps per workgroup (K), [1] forward max threads,
                                         // [2] backward K, [3] backward max threads

struct SweepParams {
    const void *x;     // forward: input; backward: saved input
    const void *go;    // backward: incoming gradient
    void *out;         // forward: output; backward: grad_x
    const void *w;
    double *partials;  // backward: [blocks_per_plane * N][C][3]
    int64_t wzp;
    uint64_t fill;
    int64_t x_plane, o_plane;  // elements per (n, c) plane
    uint32_t total;            // forward: chunks in the whole tensor; backward: chunks per plane
    uint32_t blocks;           // logical workgroups
    uint32_t blocks_per_xcd;   // ceil(blocks / 8)
    uint32_t bpp;              // backward: workgroups per plane
    int wkind, C, nd, pad;
    int S[3], O[3], L[3], wcol[3];
    uint32_t cpr;              // chunks per row of the iteration space
    uint32_t rows;             // rows per plane of the iteration space
    uint32_t CW, RPS, K;       // chunk columns / rows per step / row steps per workgroup
    uint32_t tiles;            // column tiles per row (cpr > CW)
    uint32_t threads;          // RPS * CW
    FastDiv d_CW, d_tiles, d_dim1, d_C, d_bpp;
    FastDiv d_per[3];          // divide by the padding period of each dim of x
    FastDiv d_gper[3];         // backward: same for the grad_out dims
};

// workgroups that share an XCD (blockIdx % 8, round-robin dispatch) get consecutive logical ids
__device__ __forceinline__ uint32_t xcd_remap(uint32_t blocks_per_xcd) {
    return (blockIdx.x & 7u) * blocks_per_xcd + (blockIdx.x >> 3);
}

// source index of coordinate p of a dim: -1 = fill
__device__ __forceinline__ int map1(int p, int cs, int len, int pad) { return len == 1 ? 0 : fold_index(p - cs, len, pad); }

// =====================================================================================================
// Gather forward (SSL forward of every float dtype, quantized forward)
//
// Workgroup = (plane, row band, column tile): T = RPS * CW threads cover RPS rows x CW chunk columns per
// step; a thread keeps its chunk column (column map evaluated once) and visits K rows RPS apart: all K
// loads are issued before the K stores.
// =====================================================================================================
template <int ESIZE, int V, int KMAX>
__global__ __launch_bounds__(kSweepMaxThreads) void sweep_gather_forward(const SweepParams p) {
    using R = typename raw_t<ESIZE>::type;
    constexpr int E = V / ESIZE;
    // ---- workgroup-uniform part (scalar registers) ---------------------------------------------------------
    const uint32_t bid = xcd_remap(p.blocks_per_xcd);
    if (bid >= p.blocks) return;
    const uint32_t plane = fdiv(bid, p.d_bpp);
    const uint32_t blk = bid - plane * p.bpp;
    const uint32_t band = fdiv(blk, p.d_tiles);
    const uint32_t tile = blk - band * p.tiles;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs[3];
    {
        int64_t sh[3];
        gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd, p.wcol, sh);
#pragma unroll
        for (int d = 0; d < 3; ++d) cs[d] = p.wcol[d] >= 0 ? canon_shift(sh[d], p.S[d], p.pad, p.d_per[d]) : 0;
    }
    // ---- per-thread part ---------------------------------------------------------------------------------
    const uint32_t tr = fdiv(threadIdx.x, p.d_CW);
    const uint32_t chunk = tile * p.CW + (threadIdx.x - tr * p.CW);
    if (chunk >= p.cpr) return;
    const int jo = static_cast<int>(chunk) * E;
    int mm[E];
    bool contig = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        mm[e] = map1(jo + p.L[2] + e, cs[2], p.S[2], p.pad);
        contig = contig && (mm[e] == mm[0] + e);
    }
    contig = contig && (mm[0] >= 0);

    const R fill = static_cast<R>(p.fill);
    const R *xp = static_cast<const R *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    R *op = static_cast<R *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + jo;
    const uint32_t r0 = band * (p.RPS * p.K) + tr;
    Chunk<R, E> v[KMAX];
    R *dst[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        dst[k] = nullptr;
        const uint32_t r = r0 + k * p.RPS;
        if (k < p.K && r < p.rows) {
            const uint32_t a = fdiv(r, p.d_dim1);
            const uint32_t b = r - a * static_cast<uint32_t>(p.O[1]);
            const int ra = map1(static_cast<int>(a) + p.L[0], cs[0], p.S[0], p.pad);
            const int rb = map1(static_cast<int>(b) + p.L[1], cs[1], p.S[1], p.pad);
            dst[k] = op + static_cast<int64_t>(r) * p.O[2];
            if (ra < 0 || rb < 0) {
#pragma unroll
                for (int e = 0; e < E; ++e) v[k].e[e] = fill;
            } else {
                const R *row = xp + static_cast<int64_t>(ra * p.S[1] + rb) * p.S[2];
                if (contig) {
                    v[k] = load_chunk<R, E, SWEEP_NT_LOAD>(row + mm[0]);
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) v[k].e[e] = mm[e] >= 0 ? __builtin_nontemporal_load(row + mm[e]) : fill;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (dst[k]) store_chunk<R, E>(dst[k], v[k]);
}

// =====================================================================================================
// Interpolating kernels: row loader, corner combos, float weights
// =====================================================================================================
// E (+1) consecutive mapped elements of one source row, widened to the compute type
template <typename T, int E, int CNT>
__device__ __forceinline__ void load_row(const typename T::S *__restrict__ row, bool valid, bool contig,
                                         const int (&mm)[E + 1], typename T::C (&vals)[E + 1]) {
    using S = typename T::S;
    using CT = typename T::C;
    if (!valid) {
#pragma unroll
        for (int e = 0; e <= E; ++e) vals[e] = CT(0);
        return;
    }
    if (contig) {
        const Chunk<S, E> c = load_chunk<S, E>(row + mm[0]);
#pragma unroll
        for (int e = 0; e < E; ++e) vals[e] = widen<T>(c.e[e]);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) vals[e] = mm[e] >= 0 ? widen<T>(row[mm[e]]) : CT(0);
    }
    if (CNT > E) vals[E] = mm[E] >= 0 ? widen<T>(row[mm[E]]) : CT(0);
    else vals[E] = CT(0);
}

// outer-dim corner combo k (bit r <-> +1 along real dim r < ND-1): row offset in the plane or -1
template <int ND>
__device__ __forceinline__ int combo_offset(int k, int pa, int pb, const int cs[3], const int size[3], int pad) {
    if constexpr (ND == 1) {
        return 0;
    } else if constexpr (ND == 2) {
        const int rb = map1(pb + (k & 1), cs[1], size[1], pad);
        return rb < 0 ? -1 : rb * size[2];
    } else {
        const int ra = map1(pa + (k & 1), cs[0], size[0], pad);
        const int rb = map1(pb + ((k >> 1) & 1), cs[1], size[1], pad);
        return (ra < 0 || rb < 0) ? -1 : (ra * size[1] + rb) * size[2];
    }
}

// =====================================================================================================
// Active (interpolating) forward in the sweep shape: out = interp of the 2^ND corners around (coord - floor(w))
// (shifts_kernels.h:187-205).  Same workgroup shape as the gather forward; a thread keeps its chunk column (E + 1
// mapped source columns) and visits K rows RPS apart, loading the 2^(ND-1) corner rows of each with element-aligned
// 16-byte loads.  For rows made of whole 16-byte chunks only (4- and 8-byte elements: 16-bit rows go through LDS,
// shiftnd_plane.hip / shiftnd_slide.hip, where 2-byte-aligned 16-byte global loads are slow).
// =====================================================================================================
template <typename T, int ND, int KMAX>
__global__ __launch_bounds__(kSweepMaxThreads) void sweep_active_forward(const SweepParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
    const uint32_t bid = xcd_remap(p.blocks_per_xcd);
    if (bid >= p.blocks) return;
    const uint32_t plane = fdiv(bid, p.d_bpp);
    const uint32_t blk = bid - plane * p.bpp;
    const uint32_t band = fdiv(blk, p.d_tiles);
    const uint32_t tile = blk - band * p.tiles;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0) {
            int64_t iw;
            prep_shift_forward<CT>(wv[d], true, iw, dw[p.wcol[d]]);
            cs[d] = canon_shift(iw, p.S[d], p.pad, p.d_per[d]);
        }
    const uint32_t tr = fdiv(threadIdx.x, p.d_CW);
    const uint32_t chunk = tile * p.CW + (threadIdx.x - tr * p.CW);
    if (chunk >= p.cpr) return;
    const int jo = static_cast<int>(chunk) * E;
    int mm[E + 1];
    bool contig = true;
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        mm[e] = map1(jo + p.L[2] + e, cs[2], p.S[2], p.pad);
        if (e < E) contig = contig && (mm[e] == mm[0] + e);
    }
    contig = contig && (mm[0] >= 0);
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + jo;
    const uint32_t r0 = band * (p.RPS * p.K) + tr;
    CT vals[KMAX][NC][E + 1];
    S *dst[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {  // every load first
        dst[k] = nullptr;
        const uint32_t r = r0 + k * p.RPS;
        if (k < static_cast<int>(p.K) && r < p.rows) {
            const int a = static_cast<int>(fdiv(r, p.d_dim1));
            const int b = static_cast<int>(r) - a * p.O[1];
            dst[k] = op + static_cast<int64_t>(r) * p.O[2];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int off = combo_offset<ND>(q, a + p.L[0], b + p.L[1], cs, p.S, p.pad);
                load_row<T, E, E + 1>(xp + (off < 0 ? 0 : off), off >= 0, contig, mm, vals[k][q]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (dst[k]) {
            Chunk<S, E> res;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[1 << ND];
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) v[q] = vals[k][q & (NC - 1)][e + (q >> (ND - 1))];
                res.e[e] = narrow<T>(interp_t<T, ND>(v, dw));
            }
            store_chunk<S, E>(dst[k], res);
        }
    }
}

// workgroup-wide fp64 sum for up to kSweepMaxThreads threads; result valid in thread 0
__device__ __forceinline__ double block_sum_n(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) t += scratch[w];
    return t;
}

// =====================================================================================================
// Backward: grad_x and the weight-gradient partials; iteration space = input coordinates.
// Same workgroup shape as the gather kernel: (plane, band of RPS*K input rows, column tile).
// =====================================================================================================
template <typename T, int ND, bool ACTIVE, int KMAX>
__global__ __launch_bounds__(kSweepMaxThreads) void sweep_backward(const SweepParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
    __shared__ double scratch[kSweepMaxThreads / 64];
    // ---- workgroup-uniform part -----------------------------------------------------------------------------
    const uint32_t bid = xcd_remap(p.blocks_per_xcd);
    const bool live_block = bid < p.blocks;
    const uint32_t plane = live_block ? fdiv(bid, p.d_bpp) : 0;
    const uint32_t blk = bid - plane * p.bpp;
    const uint32_t band = fdiv(blk, p.d_tiles);
    const uint32_t tile = blk - band * p.tiles;
    const uint32_t n = fdiv(plane, p.d_C);
    const int c = static_cast<int>(plane - n * static_cast<uint32_t>(p.C));
    int cs[3] = {0, 0, 0}, gs[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0) {
            int64_t iw;
            prep_shift_backward<CT>(wv[d], ACTIVE, iw, dw[p.wcol[d]]);
            cs[d] = canon_shift(iw, p.S[d], p.pad, p.d_per[d]);
            // grad_x source: SSL reads grad_out at o + shift, active at o - shift (shifts_kernels.h:287-293)
            gs[d] = canon_shift(ACTIVE ? iw : -iw, p.O[d], p.pad, p.d_gper[d]);
        }
    // ---- per-thread column state ---------------------------------------------------------------------------
    const uint32_t tr = fdiv(threadIdx.x, p.d_CW);
    const uint32_t chunk = tile * p.CW + (threadIdx.x - tr * p.CW);
    const bool live = live_block && chunk < p.cpr;
    const int ji = static_cast<int>(chunk) * E;  // input inner coordinate of element 0
    const int oj = ji - p.L[2];                   // grad_out inner coordinate of element 0 (may be outside)
    int xm[E + 1], gm[E + 1];
    unsigned inmask = 0;
    bool xcontig = true, gcontig = true;
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        xm[e] = live ? map1(ji + e, cs[2], p.S[2], p.pad) : -1;
        if (e < E) xcontig = xcontig && (xm[e] == xm[0] + e);
        const int o = oj + e;
        const bool in = (o >= 0) && (o < p.O[2]);
        if (e < E && in) inmask |= 1u << e;
        const int oc = o < 0 ? 0 : (o > p.O[2] ? p.O[2] : o);  // entries of outside elements are never used
        gm[e] = live ? map1(oc, gs[2], p.O[2], p.pad) : -1;
        if (e < E) gcontig = gcontig && (gm[e] == gm[0] + e);
    }
    const bool allin = inmask == ((1u << E) - 1u);
    xcontig = xcontig && (xm[0] >= 0);
    gcontig = gcontig && allin && (gm[0] >= 0);

    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * p.o_plane;
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane + ji;
    const uint32_t r0 = band * (p.RPS * p.K) + tr;
    double acc[3] = {0.0, 0.0, 0.0};

#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const uint32_t r = r0 + k * p.RPS;
        if (!(live && k < static_cast<int>(p.K) && r < p.rows)) continue;
        const int a = static_cast<int>(fdiv(r, p.d_dim1));
        const int b = static_cast<int>(r) - a * p.S[1];
        const int oa = a - p.L[0], ob = b - p.L[1];
        S *dst = gxp + static_cast<int64_t>(r) * p.S[2];
        Chunk<S, E> res;
        const bool rowin = (oa >= 0) && (oa < p.O[0]) && (ob >= 0) && (ob < p.O[1]);
        if (!rowin || inmask == 0) {  // outside the border window: grad_x = 0, no weight-gradient term
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = narrow<T>(CT(0));
            store_chunk<S, E>(dst, res);
            continue;
        }
        // incoming gradient at this position
        CT gval[E];
        {
            const S *grow = gp + static_cast<int64_t>(oa * p.O[1] + ob) * p.O[2];
            if (allin) {
                const Chunk<S, E> cg = load_chunk<S, E>(grow + oj);
#pragma unroll
                for (int e = 0; e < E; ++e) gval[e] = widen<T>(cg.e[e]);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) gval[e] = ((inmask >> e) & 1u) ? widen<T>(grow[oj + e]) : CT(0);
            }
        }
        // corners of x around (coord - shift) -> weight gradient
        CT xv[NC][E + 1];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int off = combo_offset<ND>(q, a, b, cs, p.S, p.pad);
            load_row<T, E, E + 1>(xp + (off < 0 ? 0 : off), off >= 0, xcontig, xm, xv[q]);
        }
        // grad_x source values
        CT gv[ACTIVE ? NC : 1][E + 1];
        Chunk<S, E> graw;
        bool gvalid = true;
        if constexpr (ACTIVE) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int off = combo_offset<ND>(q, oa, ob, gs, p.O, p.pad);
                load_row<T, E, E + 1>(gp + (off < 0 ? 0 : off), off >= 0, gcontig, gm, gv[q]);
            }
        } else {
            const int ra = map1(oa, gs[0], p.O[0], p.pad), rb = map1(ob, gs[1], p.O[1], p.pad);
            gvalid = ra >= 0 && rb >= 0;
            if (gvalid) {
                const S *srow = gp + static_cast<int64_t>(ra * p.O[1] + rb) * p.O[2];
                if (gcontig) {
                    graw = load_chunk<S, E>(srow + gm[0]);
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        graw.e[e] = (((inmask >> e) & 1u) && gm[e] >= 0) ? srow[gm[e]] : narrow<T>(CT(0));
                }
            }
        }
        // arithmetic
#pragma unroll
        for (int e = 0; e < E; ++e) {
            CT v[1 << ND], wg[3];
#pragma unroll
            for (int q = 0; q < (1 << ND); ++q) v[q] = xv[q & (NC - 1)][e + (q >> (ND - 1))];
            weight_grads_nd<ND, CT>(v, dw, wg);
            const bool in = ((inmask >> e) & 1u) != 0;
            if (in) {
#pragma unroll
                for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval[e] * wg[s]);
            }
            if constexpr (ACTIVE) {
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) v[q] = gv[q & (NC - 1)][e + (q >> (ND - 1))];
                const CT r1 = interp_t<T, ND>(v, dw);
                res.e[e] = narrow<T>(in ? r1 : CT(0));
            } else {
                res.e[e] = gvalid ? graw.e[e] : narrow<T>(CT(0));
            }
        }
        store_chunk<S, E>(dst, res);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double t = block_sum_n(acc[s], scratch);
        if (threadIdx.x == 0 && live_block)
            p.partials[((static_cast<size_t>(n) * p.bpp + blk) * p.C + c) * 3 + s] = t;
    }
}

// =====================================================================================================
// Host side
// =====================================================================================================
bool contiguous(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

int gather_vector_bytes(const Geometry &g, int esize, const void *out) {
    const int cand[3] = {16, 8, 4};
    for (int V : cand) {
        if (V < esize) continue;
        if ((g.O[2] * esize) % V != 0) continue;
        if (reinterpret_cast<uintptr_t>(out) % V != 0) continue;
        return V;
    }
    return esize;
}

void fill_common(SweepParams &p, const Geometry &g) {
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.O[d] = static_cast<int>(g.O[d]);
        p.L[d] = static_cast<int>(g.L[d]);
        p.wcol[d] = g.wcol[d];
    }
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.o_plane = g.O[0] * g.O[1] * g.O[2];
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    for (int d = 0; d < 3; ++d) {
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], p.pad)));
        p.d_gper[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.O[d], p.pad)));
    }
}

void set_grid(SweepParams &p, uint64_t blocks) {
    p.blocks = static_cast<uint32_t>(blocks);
    p.blocks_per_xcd = static_cast<uint32_t>((blocks + 7) / 8);
}

// Workgroup shape: CW chunk columns x RPS rows, T = CW * RPS <= max threads with as few idle lanes in the
// last wave as possible (e.g. 56 chunks per row -> 448 threads = 7 full waves = 8 rows).
void plan_shape(SweepParams &p, uint32_t cpr, uint32_t rows, int64_t planes, int k_want, int t_want, int k_cap) {
    const uint32_t tmax = static_cast<uint32_t>(t_want < 64 ? 64 : (t_want > kSweepMaxThreads ? kSweepMaxThreads : t_want));
    if (cpr < 1) cpr = 1;  // only reachable for ineligible geometries (workspace sizing of rows shorter than 16 bytes)
    if (rows < 1) rows = 1;
    p.cpr = cpr;
    p.rows = rows;
    p.CW = cpr < tmax ? cpr : tmax;
    p.tiles = (cpr + p.CW - 1) / p.CW;
    uint32_t best_rps = 1;
    double best = -1.0;
    const uint32_t max_rps = tmax / p.CW < rows ? tmax / p.CW : rows;
    for (uint32_t rps = 1; rps <= (max_rps ? max_rps : 1); ++rps) {
        const uint32_t t = rps * p.CW;
        const double eff = static_cast<double>(t) / (((t + 63) / 64) * 64);
        const double score = eff + 1e-4 * t;  // lane efficiency first, then larger workgroups
        if (score > best) { best = score; best_rps = rps; }
    }
    p.RPS = best_rps;
    p.threads = p.RPS * p.CW;
    uint32_t K = static_cast<uint32_t>(k_want < 1 ? 1 : (k_want > k_cap ? k_cap : k_want));
    p.K = K;
    const uint32_t bands = (rows + p.RPS * K - 1) / (p.RPS * K);
    p.bpp = bands * p.tiles;
    p.d_bpp = make_fastdiv(p.bpp);
    p.d_tiles = make_fastdiv(p.tiles);
    p.d_CW = make_fastdiv(p.CW);
    set_grid(p, static_cast<uint64_t>(planes) * p.bpp);
}

// Row steps per workgroup (the kernels' K) are fixed at the values the round-2 sweeps settled on -- 4 for the gather forward,
// 2 for the backward, 2^(3 - nd) capped at 2 .. 4 for the interpolating forward: one instantiation per form (the other K's,
// once reachable through knobs 8 / 10, were 99 kernels nothing routed to).
constexpr int kGatherK = 4, kBackwardK = 2;

template <int ESIZE, int V> void launch_gather(const SweepParams &p, hipStream_t st) {
    const dim3 grid(p.blocks_per_xcd * 8), block(p.threads);
    // (an occupancy limit through unused dynamic LDS was tried here -- fewer bytes in flight help a plain copy, tools/stream_probe
    //  X4 -- and changed nothing: 1.105 ms at every setting)
    hipLaunchKernelGGL((sweep_gather_forward<ESIZE, V, kGatherK>), grid, block, 0, st, p);
}


template <typename T, int ND, bool ACTIVE> void launch_backward_k(const SweepParams &p, hipStream_t st) {
    const dim3 grid(p.blocks_per_xcd * 8), block(p.threads);
    hipLaunchKernelGGL((sweep_backward<T, ND, ACTIVE, kBackwardK>), grid, block, 0, st, p);
}
template <typename T, bool ACTIVE> void launch_backward_nd(const SweepParams &p, hipStream_t st) {
    switch (p.nd) {
    case 1: launch_backward_k<T, 1, ACTIVE>(p, st); break;
    case 2: launch_backward_k<T, 2, ACTIVE>(p, st); break;
    default: launch_backward_k<T, 3, ACTIVE>(p, st); break;
    }
}
template <typename T> int launch_backward(const SweepParams &p, bool active, int groups, void *gw, hipStream_t st) {
    if (active) launch_backward_nd<T, true>(p, st);
    else launch_backward_nd<T, false>(p, st);
    reduce_weight_grads_of<T>(p.partials, groups, p.C, p.nd, gw, st);
    return SHIFTND_OK;
}

void plan_backward(SweepParams &p, const Geometry &g, int es) {
    fill_common(p, g);
    p.d_dim1 = make_fastdiv(static_cast<uint32_t>(g.S[1]));
    plan_shape(p, static_cast<uint32_t>(g.S[2] * es / 16), static_cast<uint32_t>(g.S[0] * g.S[1]), g.N * g.C,
               kBackwardK, g_sweep_tune[3], kBackwardK);
}

}  // namespace

bool sweep_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    (void)x;
    (void)out;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe >= (1LL << 30) || oe >= (1LL << 30)) return false;  // 32-bit in-plane offsets
    if (g.N * g.C >= (1LL << 31)) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O)) return false;
    const bool interpolating = g.active && dtype <= SHIFTND_BF16;
    const int es = dtype_size(dtype);
    if (interpolating) {  // sweep_active_forward: rows of whole 16-byte chunks; 4- / 8-byte elements, and 2-D / 3-D 16-bit tensors (the caller
                          // prefers the LDS kernels for those: this is what rows beyond their reach -- 2-D rows of more than 12 288
                          // elements, 16-bit -- take instead of the strided fallback; 1-D 16-bit rows are row_active_forward's)
        if ((es < 4 && g.nd == 1) || (g.O[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return false;
        return g.N * g.C * (oe * es / 16) < (1LL << 31) - 16;
    }
    const int V = gather_vector_bytes(g, es, out);
    const int64_t cpp = oe * es / V;
    return g.N * g.C * cpp < (1LL << 31) - 16;  // 32-bit workgroup ids (also the grid limit): one per chunk at worst
}

template <typename T> void launch_active_forward(const SweepParams &p, hipStream_t st) {
    const dim3 grid(p.blocks_per_xcd * 8), block(p.threads);
    switch (p.nd) {   // (K = the plan's: 4 rows per thread for 1-D, 2 otherwise)
    case 1:
        if constexpr (sizeof(typename T::S) >= 4) hipLaunchKernelGGL((sweep_active_forward<T, 1, 4>), grid, block, 0, st, p);
        break;
    case 2: hipLaunchKernelGGL((sweep_active_forward<T, 2, 2>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((sweep_active_forward<T, 3, 2>), grid, block, 0, st, p); break;
    }
}

int sweep_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                  void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    if (g.active && dtype <= SHIFTND_BF16) {
        note_kernel("sweep_active_forward");
        SweepParams p{};
        fill_common(p, g);
        p.x = x;
        p.out = out;
        p.w = w;
        p.wkind = wkind;
        p.d_dim1 = make_fastdiv(static_cast<uint32_t>(g.O[1]));
        const int kmax = g.nd == 1 ? 4 : 2;  // rows per thread: 2^(nd-1) corner rows of E + 1 values each stay in registers (2-D: K=2 measured best)
        plan_shape(p, static_cast<uint32_t>(g.O[2] * es / 16), static_cast<uint32_t>(g.O[0] * g.O[1]), g.N * g.C,
                   kmax, g_sweep_tune[1], kmax);
        if (dtype == SHIFTND_F32) launch_active_forward<f32_t>(p, st);
        else if (dtype == SHIFTND_F64) launch_active_forward<f64_t>(p, st);
        else if (dtype == SHIFTND_F16) launch_active_forward<f16_t>(p, st);
        else launch_active_forward<bf16_t>(p, st);
        return SHIFTND_OK;
    }
    const int V = gather_vector_bytes(g, es, out);
    note_kernel("sweep_gather_forward");
    SweepParams p{};
    fill_common(p, g);
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = fill_bits;
    p.d_dim1 = make_fastdiv(static_cast<uint32_t>(g.O[1]));
    plan_shape(p, static_cast<uint32_t>(g.O[2] * es / V), static_cast<uint32_t>(g.O[0] * g.O[1]), g.N * g.C,
               kGatherK, g_sweep_tune[1], kGatherK);
#define SHIFTND_GATHER_CASE(ES, VV) \
    if (es == ES && V == VV) { launch_gather<ES, VV>(p, st); return SHIFTND_OK; }
    SHIFTND_GATHER_CASE(1, 16) SHIFTND_GATHER_CASE(1, 8) SHIFTND_GATHER_CASE(1, 4) SHIFTND_GATHER_CASE(1, 1)
    SHIFTND_GATHER_CASE(2, 16) SHIFTND_GATHER_CASE(2, 8) SHIFTND_GATHER_CASE(2, 4) SHIFTND_GATHER_CASE(2, 2)
    SHIFTND_GATHER_CASE(4, 16) SHIFTND_GATHER_CASE(4, 8) SHIFTND_GATHER_CASE(4, 4)
    SHIFTND_GATHER_CASE(8, 16) SHIFTND_GATHER_CASE(8, 8)
#undef SHIFTND_GATHER_CASE
    return SHIFTND_ERR_UNSUPPORTED_DTYPE;
}

bool sweep_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    (void)go;
    (void)x;
    if (dtype > SHIFTND_BF16) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe >= (1LL << 30) || oe >= (1LL << 30)) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O) || !contiguous(g.gs, g.N, g.C, g.S))
        return false;
    const int es = dtype_size(dtype);
    if ((g.S[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(gx) % 16 != 0) return false;
    return g.N * g.C * (xe * es / 16) < (1LL << 31) - 16;
}

size_t sweep_backward_workspace(const Geometry &g, int dtype) {
    SweepParams p{};
    plan_backward(p, g, dtype_size(dtype));
    return static_cast<size_t>(g.N) * p.bpp * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

int sweep_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st) {
    SweepParams p{};
    note_kernel("sweep_backward");
    plan_backward(p, g, dtype_size(dtype));
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    const int groups = static_cast<int>(g.N) * static_cast<int>(p.bpp);
    switch (dtype) {
    case SHIFTND_F32: return launch_backward<f32_t>(p, g.active != 0, groups, gw, st);
    case SHIFTND_F64: return launch_backward<f64_t>(p, g.active != 0, groups, gw, st);
    case SHIFTND_F16: return launch_backward<f16_t>(p, g.active != 0, groups, gw, st);
    default: return launch_backward<bf16_t>(p, g.active != 0, groups, gw, st);
    }
}

void sweep_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 4) g_sweep_tune[knob] = value;
}

// host mirror of the per-element map, for tests: source index of coordinate p (or -1)
int sweep_debug_map(int64_t p, int64_t shift, int64_t len, int pad) {
    if (len == 1) return 0;
    const FastDiv dper = make_fastdiv(static_cast<uint32_t>(map_period(static_cast<int>(len), pad)));
    return fold_index(static_cast<int>(p) - canon_shift(shift, static_cast<int>(len), pad, dper), static_cast<int>(len), pad);
}

}  // namespace shiftnd
