// shiftnd_small.hip -- whole small planes through LDS, gfx950 (MI355X): the interpolating forward and the backward
// pass of contiguous problems whose rows are NOT whole 16-byte pieces (a 14 x 14 or 7 x 7 fp32 plane: 56- / 28-byte
// rows) -- the late stages of a ResNet-shaped network, which the row-chunk kernels (16-byte rows) cannot take and the
// one-thread-per-element fallback served at 0.15-0.7 TB/s (N128 C512 14x14 fp32: backward 0.32 ms, N128 C1024 7x7:
// 0.51 ms).
//
// A plane of <= a few thousand elements fits LDS many times over, and the shift is one number per channel and dim, so
//   * one workgroup = one channel x `ppr * rpw` batch entries, in `rpw` rounds of `ppr` planes: the per-channel index
//     maps (build_maps) are built once per workgroup;
//   * per round the planes of the saved input (and of the incoming gradient) are copied to LDS element by element with
//     coalesced loads (a plane is one contiguous run of memory, at element alignment only);
//   * a thread owns flat (plane, element) indices idx = tid + 256 k: every lane is busy whatever the plane size; the
//     element's coordinates come from two multiply-shift divisions, its corner rows / columns from the maps, the corner
//     values from LDS; stores are coalesced element stores;
//   * backward: fp64 weight-gradient sums per thread, one block sum per workgroup into the [group][C][3] partial sums
//     that reduce_weight_grads finishes (deterministic).
// Arithmetic is the shared one (interp_t, weight_grads_nd, prep_shift_*): forward and grad_x are bit-identical to the
// other kernel families in fp32 / fp64, within 1 ulp of the 16-bit type for fp16 / bf16 interpolation.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/): forward kernels/shifts_kernels.h:156-220, backward
// :222-327, interpolation kernels/interpolation.h:3-61, weight preparation cpu/shifts_cpu.cpp:223-224, :242-244.
// Roofline: HBM; forward 2 s bytes per element, backward 3 s.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

// bytes of the index maps in front of the element buffers, rounded to 16 (the buffers hold up to 8-byte elements)
__host__ __device__ inline size_t map_bytes16(int map_entries, bool two_sets) {
    return (static_cast<size_t>(map_entries) * (two_sets ? 2 : 1) * sizeof(int) + 15) & ~static_cast<size_t>(15);
}

struct SmallParams {
    const void *x;       // forward: input; backward: saved input
    const void *go;      // backward: incoming gradient
    void *out;           // forward: output; backward: grad_x
    const void *w;
    double *partials;    // backward: [groups][C][3]
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int PE;              // elements per plane
    int ppr, rpw;        // planes per round, rounds per workgroup
    int map_entries;     // S0 + S1 + S2 + 3
    unsigned xcd_blocks;
    FastDiv d_S2, d_S12, d_PE, d_C;
    FastDiv d_per[3];
};

template <typename T, int ND, bool ACTIVE, bool BACKWARD>
__global__ __launch_bounds__(kThreads) void small_plane_kernel(const SmallParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int NC = 1 << ND;
    static_assert(ACTIVE || BACKWARD, "the sparse-shift forward is served by the gather kernels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double scratch[kThreads / 64];
    int *maps = reinterpret_cast<int *>(smem);
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2], PE = p.PE;
    const int *m0 = maps, *m1 = m0 + S0 + 1, *m2 = m1 + S1 + 1;
    int *gmaps = maps + p.map_entries;
    const int *g0 = gmaps, *g1 = g0 + S0 + 1, *g2 = g1 + S1 + 1;
    S *xbuf = reinterpret_cast<S *>(smem + map_bytes16(p.map_entries, BACKWARD));
    S *gbuf = xbuf + static_cast<size_t>(p.ppr) * PE;   // (backward only)

    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int grp = fdiv(bid, p.d_C), c = static_cast<int>(bid) - grp * p.C;
    const int n0 = grp * p.ppr * p.rpw, nw = min(p.ppr * p.rpw, p.N - n0);

    // ---- per-channel shift and maps ------------------------------------------------------------------------------------
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.wcol[d] >= 0) {
            if constexpr (BACKWARD) prep_shift_backward<CT>(wv[d], ACTIVE, sh[d], dw[p.wcol[d]]);
            else prep_shift_forward<CT>(wv[d], true, sh[d], dw[p.wcol[d]]);
        }
    }
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    // grad_x source: the sparse shift reads grad_out at o + shift, the active one at o - shift (shifts_kernels.h:287-293)
    if constexpr (BACKWARD) build_maps(gmaps, p.S, sh, ACTIVE ? -1 : +1, p.pad, p.d_per);

    const int64_t nstride = static_cast<int64_t>(p.C) * PE;
    const S *xb = static_cast<const S *>(p.x) + (static_cast<int64_t>(n0) * p.C + c) * PE;
    const S *gb = BACKWARD ? static_cast<const S *>(p.go) + (static_cast<int64_t>(n0) * p.C + c) * PE : xb;
    S *ob = static_cast<S *>(p.out) + (static_cast<int64_t>(n0) * p.C + c) * PE;
    double acc[3] = {0.0, 0.0, 0.0};

    // corner values of one element: rows / columns through the maps, values from the staged plane (zero where the
    // padding map says so).  Corner order (shifts_kernels.h:58-103): bit r = +1 along the r-th spatial dim of the
    // tensor, i.e. along the normalised dims [0][1][2] in 3-D, [1][2] in 2-D, [2] in 1-D.
    auto corners = [&](const S *plane, const int *q0, const int *q1, const int *q2, int a, int b, int cc, CT (&v)[NC]) {
        if constexpr (ND == 1) {
            const int c0 = q2[cc], c1 = q2[cc + 1];
            v[0] = c0 >= 0 ? widen<T>(plane[c0]) : CT(0);
            v[1] = c1 >= 0 ? widen<T>(plane[c1]) : CT(0);
        } else if constexpr (ND == 2) {
            const int r0 = q1[b], r1 = q1[b + 1], c0 = q2[cc], c1 = q2[cc + 1];
            const int rr[2] = {r0, r1}, ccs[2] = {c0, c1};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rr[q & 1], col = ccs[q >> 1];
                const bool ok = r >= 0 && col >= 0;
                const S raw = plane[ok ? r * S2 + col : 0];
                v[q] = ok ? widen<T>(raw) : CT(0);
            }
        } else {
            const int d0 = q0[a], d1 = q0[a + 1], r0 = q1[b], r1 = q1[b + 1], c0 = q2[cc], c1 = q2[cc + 1];
            const int dd[2] = {d0, d1}, rr[2] = {r0, r1}, ccs[2] = {c0, c1};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int dep = dd[q & 1], r = rr[(q >> 1) & 1], col = ccs[q >> 2];
                const bool ok = r >= 0 && col >= 0 && dep >= 0;
                const S raw = plane[ok ? (dep * S1 + r) * S2 + col : 0];
                v[q] = ok ? widen<T>(raw) : CT(0);
            }
        }
    };

    for (int r = 0; r * p.ppr < nw; ++r) {
        const int np = min(p.ppr, nw - r * p.ppr), total = np * PE;
        const int64_t rbase = static_cast<int64_t>(r) * p.ppr * nstride;
        __syncthreads();  // the maps are complete / the previous round is done with the buffers
        for (int idx = threadIdx.x; idx < total; idx += kThreads) {
            const int pl = fdiv(idx, p.d_PE), o = idx - pl * PE;
            xbuf[idx] = xb[rbase + pl * nstride + o];
            if constexpr (BACKWARD) gbuf[idx] = gb[rbase + pl * nstride + o];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < total; idx += kThreads) {
            const int pl = fdiv(idx, p.d_PE), o = idx - pl * PE;
            const int a = fdiv(o, p.d_S12), rem = o - a * (S1 * S2);
            const int b = fdiv(rem, p.d_S2), cc = rem - b * S2;
            const S *xpl = xbuf + pl * PE;
            S res;
            if constexpr (BACKWARD) {
                const S *gpl = gbuf + pl * PE;
                CT v[NC], wg[3];
                corners(xpl, m0, m1, m2, a, b, cc, v);
                const CT gval = widen<T>(gbuf[idx]);
                weight_grads_nd<ND, CT>(v, dw, wg);
#pragma unroll
                for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval * wg[s]);
                if constexpr (ACTIVE) {
                    corners(gpl, g0, g1, g2, a, b, cc, v);
                    res = narrow<T>(interp_t<T, ND>(v, dw));
                } else {
                    const int ra = ND == 3 ? g0[a] : 0, rb = ND >= 2 ? g1[b] : 0, rc = g2[cc];
                    const bool ok = ra >= 0 && rb >= 0 && rc >= 0;
                    const S raw = gpl[ok ? (ra * S1 + rb) * S2 + rc : 0];
                    res = ok ? raw : narrow<T>(CT(0));   // pure copy: the bit pattern is kept
                }
            } else {
                CT v[NC];
                corners(xpl, m0, m1, m2, a, b, cc, v);
                res = narrow<T>(interp_t<T, ND>(v, dw));
            }
            ob[rbase + pl * nstride + o] = res;
        }
    }

    if constexpr (BACKWARD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tsum = block_sum(acc[k], scratch);
            if (threadIdx.x == 0) p.partials[(static_cast<size_t>(grp) * p.C + c) * 3 + k] = tsum;
        }
    }
}

// =====================================================================================================
// Row bands: the same idea for planes that do not fit (1-D / 2-D, rows of any length that is not a whole number of
// 16-byte pieces: 113 x 113, 225 x 225, 299 x 299 ...).  A workgroup owns one channel and a run of (batch entry, band of
// BR rows) units; per unit it stages the BR + 1 SOURCE rows of the band (row slot j = map1[b0 + j]: the "+ 1 along H"
// corner row of output row b is the first corner row of row b + 1; rows the padding map declares fill are staged as
// zeros) and, for the backward, the BR (+ 1) gradient rows its grad_x reads; the incoming gradient at the output
// position is read straight from memory (coalesced, every element once).
// =====================================================================================================
struct BandParams {
    const void *x, *go;
    void *out;
    const void *w;
    double *partials;
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int BR, bands, units, upw;   // rows per band, bands per plane, N * bands, units per workgroup
    int map_entries;
    unsigned xcd_blocks;
    FastDiv d_S2, d_bands, d_C;
    FastDiv d_per[3];
};

template <typename T, int ND, bool ACTIVE, bool BACKWARD>
__global__ __launch_bounds__(kThreads) void band_plane_kernel(const BandParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int NC = 1 << ND;
    static_assert(ND == 1 || ND == 2, "one or two spatial dims");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double scratch[kThreads / 64];
    int *maps = reinterpret_cast<int *>(smem);
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2], BR = p.BR;
    const int *m1 = maps + S0 + 1, *m2 = m1 + S1 + 1;
    int *gmaps = maps + p.map_entries;
    const int *g1 = gmaps + S0 + 1, *g2 = g1 + S1 + 1;
    S *xrows = reinterpret_cast<S *>(smem + map_bytes16(p.map_entries, BACKWARD));
    S *grows = xrows + static_cast<size_t>(ND == 1 ? 1 : BR + 1) * S2;   // (backward only; 1-D: the one row)

    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int grp = fdiv(bid, p.d_C), c = static_cast<int>(bid) - grp * p.C;
    const int u0 = grp * p.upw, nu = min(p.upw, p.units - u0);

    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.wcol[d] >= 0) {
            if constexpr (BACKWARD) prep_shift_backward<CT>(wv[d], ACTIVE, sh[d], dw[p.wcol[d]]);
            else prep_shift_forward<CT>(wv[d], true, sh[d], dw[p.wcol[d]]);
        }
    }
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    if constexpr (BACKWARD) build_maps(gmaps, p.S, sh, ACTIVE ? -1 : +1, p.pad, p.d_per);

    const int64_t plane = static_cast<int64_t>(S1) * S2;
    double acc[3] = {0.0, 0.0, 0.0};
    const S zero = narrow<T>(CT(0));
    for (int k = 0; k < nu; ++k) {
        const int u = u0 + k;
        const int n = fdiv(u, p.d_bands), band = u - n * p.bands;
        const int b0 = band * BR, nb = min(BR, S1 - b0);
        const int64_t pbase = (static_cast<int64_t>(n) * p.C + c) * plane;
        const S *xp = static_cast<const S *>(p.x) + pbase;
        const S *gp = BACKWARD ? static_cast<const S *>(p.go) + pbase : xp;
        S *op = static_cast<S *>(p.out) + pbase;
        __syncthreads();  // the maps are complete / the previous unit is done with the rows
        for (int idx = threadIdx.x; idx < (ND == 1 ? 1 : nb + 1) * S2; idx += kThreads) {
            const int j = fdiv(idx, p.d_S2), cc = idx - j * S2;
            const int rx = m1[b0 + j];
            xrows[idx] = rx >= 0 ? xp[static_cast<int64_t>(rx) * S2 + cc] : zero;
            if constexpr (BACKWARD) {
                const int rg = (ACTIVE || j < nb) ? g1[b0 + j] : -1;
                grows[idx] = rg >= 0 ? gp[static_cast<int64_t>(rg) * S2 + cc] : zero;
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * S2; idx += kThreads) {
            const int j = fdiv(idx, p.d_S2), cc = idx - j * S2;
            const int64_t o = static_cast<int64_t>(b0 + j) * S2 + cc;
            // corner order: bit 0 = +1 along H (2-D) / along L (1-D), bit 1 = +1 along W (2-D)
            auto corners = [&](const S *rows, const int *q2, CT (&v)[NC]) {
                const int c0 = q2[cc], c1 = q2[cc + 1];
                if constexpr (ND == 1) {
                    v[0] = c0 >= 0 ? widen<T>(rows[c0]) : CT(0);
                    v[1] = c1 >= 0 ? widen<T>(rows[c1]) : CT(0);
                } else {
                    const S *r0 = rows + j * S2, *r1 = r0 + S2;
                    v[0] = c0 >= 0 ? widen<T>(r0[c0]) : CT(0);
                    v[1] = c0 >= 0 ? widen<T>(r1[c0]) : CT(0);
                    v[2] = c1 >= 0 ? widen<T>(r0[c1]) : CT(0);
                    v[3] = c1 >= 0 ? widen<T>(r1[c1]) : CT(0);
                }
            };
            S res;
            CT v[NC];
            if constexpr (BACKWARD) {
                CT wg[3];
                corners(xrows, m2, v);
                const CT gval = widen<T>(gp[o]);
                weight_grads_nd<ND, CT>(v, dw, wg);
#pragma unroll
                for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval * wg[s]);
                if constexpr (ACTIVE) {
                    corners(grows, g2, v);
                    res = narrow<T>(interp_t<T, ND>(v, dw));
                } else {
                    const int rc = g2[cc];
                    res = rc >= 0 ? grows[j * S2 + rc] : zero;   // pure copy: the bit pattern is kept
                }
            } else {
                corners(xrows, m2, v);
                res = narrow<T>(interp_t<T, ND>(v, dw));
            }
            op[o] = res;
        }
    }
    if constexpr (BACKWARD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tsum = block_sum(acc[k], scratch);
            if (threadIdx.x == 0) p.partials[(static_cast<size_t>(grp) * p.C + c) * 3 + k] = tsum;
        }
    }
}

// Sparse-shift / quantized forward over the same row bands (any element size): out[b][c] = x[map1[b]][map2[c]] or the
// fill value.  For rows the chunk kernels move element by element (rows that are not whole 16-byte pieces: 0.7 TB/s for
// 1-byte elements, 2.9 TB/s for fp32 on 225-wide rows).
struct BandGatherParams {
    const void *x;
    void *out;
    const void *w;
    int64_t wzp;
    uint64_t fill;
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int BR, bands, units, upw;
    int map_entries;
    unsigned xcd_blocks;
    FastDiv d_S2, d_bands, d_C;
    FastDiv d_per[3];
};

template <int ESIZE>
__global__ __launch_bounds__(kThreads) void band_gather_kernel(const BandGatherParams p) {
    using R = typename raw_t<ESIZE>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *maps = reinterpret_cast<int *>(smem);
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2], BR = p.BR;
    const int *m1 = maps + S0 + 1, *m2 = m1 + S1 + 1;
    R *xrows = reinterpret_cast<R *>(smem + map_bytes16(p.map_entries, false));

    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int grp = fdiv(bid, p.d_C), c = static_cast<int>(bid) - grp * p.C;
    const int u0 = grp * p.upw, nu = min(p.upw, p.units - u0);
    int64_t sh[3];
    gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd, p.wcol, sh);
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d] = p.wcol[d] >= 0 ? sh[d] : 0;
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    const R fill = static_cast<R>(p.fill);
    const int64_t plane = static_cast<int64_t>(S1) * S2;
    for (int k = 0; k < nu; ++k) {
        const int u = u0 + k;
        const int n = fdiv(u, p.d_bands), band = u - n * p.bands;
        const int b0 = band * BR, nb = min(BR, S1 - b0);
        const int64_t pbase = (static_cast<int64_t>(n) * p.C + c) * plane;
        const R *xp = static_cast<const R *>(p.x) + pbase;
        R *op = static_cast<R *>(p.out) + pbase;
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * S2; idx += kThreads) {
            const int j = fdiv(idx, p.d_S2), cc = idx - j * S2;
            const int rx = m1[b0 + j];
            xrows[idx] = rx >= 0 ? xp[static_cast<int64_t>(rx) * S2 + cc] : fill;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * S2; idx += kThreads) {
            const int j = fdiv(idx, p.d_S2), cc = idx - j * S2;
            const int rc = m2[cc];
            op[static_cast<int64_t>(b0) * S2 + idx] = rc >= 0 ? xrows[j * S2 + rc] : fill;
        }
    }
}

struct BandPlan {
    int BR, bands, units, upw, groups, map_entries;
    size_t lds;
    unsigned grid;
    bool ok;
};

// (Round 2's table form of this kernel for planes of at most 1024 elements -- the 14 x 14 / 7 x 7 stages -- and the 1-D / 2-D
// instantiations went with round 5: those shapes are the flat-stream kernels', shiftnd_flat.hip.)

struct SmallPlan {
    int ppr, rpw, groups, map_entries;
    bool table;   // small_table_kernel (planes of at most kTablePlane elements)
    size_t lds;
    unsigned grid;
    bool ok;
};

thread_local int g_small_tune[3] = {1, 0, 0};  // [0] enabled, [1] planes per round (0 = automatic), [2] rounds per workgroup

bool contiguous5s(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

constexpr int kSmallMaxPlaneBytes = 16 * 1024;   // per tensor
constexpr int kSmallLds = 32 * 1024;             // planes of a round, at most
constexpr int kSmallRoundElems = 1536;           // elements of a round, about
constexpr int kSmallWgs = 4096;                  // workgroups wanted

SmallPlan small_plan(const Geometry &g, int es, bool backward) {
    SmallPlan pl{};
    pl.ok = false;
    pl.table = false;
    if (g.nd != 3) return pl;   // (round 5: 1-D / 2-D planes with ragged rows are the flat-stream kernels', shiftnd_flat.hip)
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t pe = g.S[0] * g.S[1] * g.S[2];
    if (pe < 1 || pe * es > kSmallMaxPlaneBytes) return pl;
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return pl;
    pl.map_entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    const size_t map_bytes = map_bytes16(pl.map_entries, backward);
    if (map_bytes > 16 * 1024) return pl;
    const int64_t per_plane = pe * es * (backward ? 2 : 1);
    // planes per round: ~1536 elements (six per thread) measured best -- N128 C512 14x14 fp32 backward 0.091 ms with 31
    // planes (48 KiB) per round, 0.056 ms with 8; N128 C1024 7x7 0.075 ms with 64, 0.037 ms with 16 -- small rounds keep
    // many workgroups per CU resident, and those overlap each other's copy-in and compute phases
    int64_t ppr = g_small_tune[1] > 0 ? g_small_tune[1] : kSmallRoundElems / pe;
    if (ppr * per_plane > kSmallLds) ppr = kSmallLds / per_plane;
    if (ppr < 1) ppr = 1;
    if (ppr > g.N) ppr = g.N;
    // workgroups per channel: enough for ~kSmallWgs workgroups, the batch split evenly between them (a ragged split --
    // 124 + 4 planes -- leaves half of the workgroups idle), then even rounds inside a workgroup
    int64_t rpw;
    if (g_small_tune[2] > 0) {
        rpw = g_small_tune[2];
    } else {
        const int64_t rounds = (g.N + ppr - 1) / ppr;
        int64_t groups = (kSmallWgs + g.C - 1) / g.C;
        if (groups > rounds) groups = rounds;
        if (groups < 1) groups = 1;
        const int64_t per_wg = (g.N + groups - 1) / groups;
        rpw = (per_wg + ppr - 1) / ppr;
        if (g_small_tune[1] <= 0) ppr = (per_wg + rpw - 1) / rpw;
    }
    if (ppr * rpw > g.N) rpw = (g.N + ppr - 1) / ppr;
    pl.ppr = static_cast<int>(ppr);
    pl.rpw = static_cast<int>(rpw);
    pl.groups = static_cast<int>((g.N + ppr * rpw - 1) / (ppr * rpw));
    pl.lds = map_bytes + static_cast<size_t>(ppr) * static_cast<size_t>(per_plane);
    if (pl.lds > 60 * 1024) return pl;
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

void fill_small(SmallParams &p, const Geometry &g, const SmallPlan &pl) {
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.PE = p.S[0] * p.S[1] * p.S[2];
    p.ppr = pl.ppr;
    p.rpw = pl.rpw;
    p.map_entries = pl.map_entries;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(p.S[2]));
    p.d_S12 = make_fastdiv(static_cast<uint32_t>(p.S[1] * p.S[2]));
    p.d_PE = make_fastdiv(static_cast<uint32_t>(p.PE));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
}

template <typename T, bool BACKWARD>
void launch_small(const SmallParams &p, const SmallPlan &pl, bool active, hipStream_t st) {
    const dim3 grid(pl.grid), block(kThreads);
#define SHIFTND_SMALL_K(KERNEL, NDV) \
    if constexpr (BACKWARD) { \
        if (active) hipLaunchKernelGGL((KERNEL<T, NDV, true, true>), grid, block, pl.lds, st, p); \
        else hipLaunchKernelGGL((KERNEL<T, NDV, false, true>), grid, block, pl.lds, st, p); \
    } else { \
        hipLaunchKernelGGL((KERNEL<T, NDV, true, false>), grid, block, pl.lds, st, p); \
    }
    SHIFTND_SMALL_K(small_plane_kernel, 3)   // (3-D volumes only: small_plan)
#undef SHIFTND_SMALL_K
}

BandPlan band_plan(const Geometry &g, int es, bool backward) {
    BandPlan pl{};
    pl.ok = false;
    if (g.nd != 1 && g.nd != 2) return pl;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t W = g.S[2], H = g.S[1];
    if (W < 1 || H < 1 || g.S[0] != 1 || H * W >= (1LL << 30)) return pl;
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return pl;
    pl.map_entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    const size_t map_bytes = map_bytes16(pl.map_entries, backward);
    if (map_bytes > 24 * 1024) return pl;
    const int64_t row_bytes = W * es * (backward ? 2 : 1);
    // rows per band: ~1536 elements per unit (see small_plan), within 24 KiB of staged rows
    int64_t br = g_small_tune[1] > 0 ? g_small_tune[1] : (kSmallRoundElems + W - 1) / W;
    while (br > 1 && (br + 1) * row_bytes > 24 * 1024) --br;
    if (br > H) br = H;
    const int64_t staged = g.nd == 1 ? 1 : br + 1;   // rows in LDS per tensor
    if (br < 1 || staged * row_bytes > 40 * 1024) return pl;
    pl.BR = static_cast<int>(br);
    pl.bands = static_cast<int>((H + br - 1) / br);
    const int64_t units = g.N * pl.bands;
    if (units >= (1LL << 30)) return pl;
    pl.units = static_cast<int>(units);
    int64_t groups = (kSmallWgs + g.C - 1) / g.C;
    if (groups > units) groups = units;
    if (groups < 1) groups = 1;
    pl.upw = static_cast<int>((units + groups - 1) / groups);
    pl.groups = static_cast<int>((units + pl.upw - 1) / pl.upw);
    pl.lds = map_bytes + static_cast<size_t>(staged) * static_cast<size_t>(row_bytes);
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

void fill_band(BandParams &p, const Geometry &g, const BandPlan &pl) {
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.BR = pl.BR;
    p.bands = pl.bands;
    p.units = pl.units;
    p.upw = pl.upw;
    p.map_entries = pl.map_entries;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(p.S[2]));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
}

template <typename T, bool BACKWARD>
void launch_band(const BandParams &p, const BandPlan &pl, bool active, hipStream_t st) {
    const dim3 grid(pl.grid), block(kThreads);
#define SHIFTND_BAND(NDV) \
    if constexpr (BACKWARD) { \
        if (active) hipLaunchKernelGGL((band_plane_kernel<T, NDV, true, true>), grid, block, pl.lds, st, p); \
        else hipLaunchKernelGGL((band_plane_kernel<T, NDV, false, true>), grid, block, pl.lds, st, p); \
    } else { \
        hipLaunchKernelGGL((band_plane_kernel<T, NDV, true, false>), grid, block, pl.lds, st, p); \
    }
    if (p.nd == 1) { SHIFTND_BAND(1) }
    else { SHIFTND_BAND(2) }
#undef SHIFTND_BAND
}

BandPlan band_gather_plan(const Geometry &g, int es) {
    BandPlan pl{};
    pl.ok = false;
    if (g.nd != 1 && g.nd != 2) return pl;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t W = g.S[2], H = g.S[1];
    if (W < 1 || H < 1 || g.S[0] != 1 || H * W >= (1LL << 30)) return pl;
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return pl;
    pl.map_entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    const size_t map_bytes = map_bytes16(pl.map_entries, false);
    if (map_bytes > 24 * 1024) return pl;
    const int64_t row_bytes = W * es;
    int64_t br = g_small_tune[1] > 0 ? g_small_tune[1] : (2 * kSmallRoundElems + W - 1) / W;
    while (br > 1 && br * row_bytes > 24 * 1024) --br;
    if (br > H) br = H;
    if (br < 1 || br * row_bytes > 40 * 1024) return pl;
    pl.BR = static_cast<int>(br);
    pl.bands = static_cast<int>((H + br - 1) / br);
    const int64_t units = g.N * pl.bands;
    if (units >= (1LL << 30)) return pl;
    pl.units = static_cast<int>(units);
    int64_t groups = (kSmallWgs + g.C - 1) / g.C;
    if (groups > units) groups = units;
    if (groups < 1) groups = 1;
    pl.upw = static_cast<int>((units + groups - 1) / groups);
    pl.groups = static_cast<int>((units + pl.upw - 1) / pl.upw);
    pl.lds = map_bytes + static_cast<size_t>(br) * static_cast<size_t>(row_bytes);
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

}  // namespace

void small_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 3) g_small_tune[knob] = value;
}

// interpolating forward of contiguous float tensors, no crop, planes of at most 16 KiB
bool small_forward_eligible(const Geometry &g, int dtype) {
    if (!g_small_tune[0] || dtype > SHIFTND_BF16 || !g.active) return false;
    if (!contiguous5s(g.xs, g.N, g.C, g.S) || !contiguous5s(g.os, g.N, g.C, g.O)) return false;
    return small_plan(g, dtype_size(dtype), false).ok || band_plan(g, dtype_size(dtype), false).ok;
}

int small_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st) {
    const SmallPlan pl = small_plan(g, dtype_size(dtype), false);
    if (!pl.ok) {   // the plane does not fit: row bands
        const BandPlan bp = band_plan(g, dtype_size(dtype), false);
        if (!bp.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
        BandParams b{};
        b.x = x;
        b.out = out;
        b.w = w;
        b.wkind = dtype;
        fill_band(b, g, bp);
        note_kernel("band_plane_forward");
        switch (dtype) {
        case SHIFTND_F32: launch_band<f32_t, false>(b, bp, true, st); break;
        case SHIFTND_F64: launch_band<f64_t, false>(b, bp, true, st); break;
        case SHIFTND_F16: launch_band<f16_t, false>(b, bp, true, st); break;
        default: launch_band<bf16_t, false>(b, bp, true, st); break;
        }
        return SHIFTND_OK;
    }
    SmallParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = dtype;
    fill_small(p, g, pl);
    note_kernel("small_plane_forward");
    switch (dtype) {
    case SHIFTND_F32: launch_small<f32_t, false>(p, pl, true, st); break;
    case SHIFTND_F64: launch_small<f64_t, false>(p, pl, true, st); break;
    case SHIFTND_F16: launch_small<f16_t, false>(p, pl, true, st); break;
    default: launch_small<bf16_t, false>(p, pl, true, st); break;
    }
    return SHIFTND_OK;
}

bool small_backward_eligible(const Geometry &g, int dtype) {
    if (!g_small_tune[0] || dtype > SHIFTND_BF16) return false;
    if (!contiguous5s(g.xs, g.N, g.C, g.S) || !contiguous5s(g.os, g.N, g.C, g.O) || !contiguous5s(g.gs, g.N, g.C, g.S)) return false;
    return small_plan(g, dtype_size(dtype), true).ok || band_plan(g, dtype_size(dtype), true).ok;
}

size_t small_backward_workspace(const Geometry &g, int dtype) {
    if (dtype > SHIFTND_BF16) return 0;
    const SmallPlan pl = small_plan(g, dtype_size(dtype), true);
    if (pl.ok) return static_cast<size_t>(pl.groups) * static_cast<size_t>(g.C) * 3 * sizeof(double);
    const BandPlan bp = band_plan(g, dtype_size(dtype), true);
    return bp.ok ? static_cast<size_t>(bp.groups) * static_cast<size_t>(g.C) * 3 * sizeof(double) : 0;
}

template <typename T>
static void band_backward_t(const BandParams &p, const BandPlan &pl, bool active, void *gw, hipStream_t st) {
    launch_band<T, true>(p, pl, active, st);
    reduce_weight_grads_of<T>(p.partials, pl.groups, p.C, p.nd, gw, st);
}

template <typename T>
static void small_backward_t(const SmallParams &p, const SmallPlan &pl, bool active, void *gw, hipStream_t st) {
    launch_small<T, true>(p, pl, active, st);
    reduce_weight_grads_of<T>(p.partials, pl.groups, p.C, p.nd, gw, st);
}

int small_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st) {
    const SmallPlan pl = small_plan(g, dtype_size(dtype), true);
    if (!pl.ok) {   // the planes do not fit: row bands
        const BandPlan bp = band_plan(g, dtype_size(dtype), true);
        if (!bp.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
        BandParams b{};
        b.x = x;
        b.go = go;
        b.out = gx;
        b.w = w;
        b.wkind = dtype;
        b.partials = static_cast<double *>(workspace);
        fill_band(b, g, bp);
        note_kernel("band_plane_backward");
        switch (dtype) {
        case SHIFTND_F32: band_backward_t<f32_t>(b, bp, g.active != 0, gw, st); break;
        case SHIFTND_F64: band_backward_t<f64_t>(b, bp, g.active != 0, gw, st); break;
        case SHIFTND_F16: band_backward_t<f16_t>(b, bp, g.active != 0, gw, st); break;
        default: band_backward_t<bf16_t>(b, bp, g.active != 0, gw, st); break;
        }
        return SHIFTND_OK;
    }
    SmallParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_small(p, g, pl);
    note_kernel("small_plane_backward");
    switch (dtype) {
    case SHIFTND_F32: small_backward_t<f32_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_F64: small_backward_t<f64_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_F16: small_backward_t<f16_t>(p, pl, g.active != 0, gw, st); break;
    default: small_backward_t<bf16_t>(p, pl, g.active != 0, gw, st); break;
    }
    return SHIFTND_OK;
}

// sparse-shift / quantized forward of contiguous 1-D / 2-D tensors, no crop, whose rows are not whole 16-byte pieces
bool band_gather_forward_eligible(const Geometry &g, int dtype) {
    if (!g_small_tune[0] || (g.active && dtype <= SHIFTND_BF16)) return false;
    const int es = dtype_size(dtype);
    if ((g.S[2] * es) % 16 == 0) return false;   // whole pieces: the chunk kernels
    if (!contiguous5s(g.xs, g.N, g.C, g.S) || !contiguous5s(g.os, g.N, g.C, g.O)) return false;
    return band_gather_plan(g, es).ok;
}

int band_gather_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                        void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const BandPlan pl = band_gather_plan(g, es);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    BandGatherParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = fill_bits;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.BR = pl.BR;
    p.bands = pl.bands;
    p.units = pl.units;
    p.upw = pl.upw;
    p.map_entries = pl.map_entries;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(p.S[2]));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    note_kernel("band_gather_forward");
    const dim3 grid(pl.grid), block(kThreads);
    switch (es) {
    case 1: hipLaunchKernelGGL(band_gather_kernel<1>, grid, block, pl.lds, st, p); break;
    case 2: hipLaunchKernelGGL(band_gather_kernel<2>, grid, block, pl.lds, st, p); break;
    case 4: hipLaunchKernelGGL(band_gather_kernel<4>, grid, block, pl.lds, st, p); break;
    default: hipLaunchKernelGGL(band_gather_kernel<8>, grid, block, pl.lds, st, p); break;
    }
    return SHIFTND_OK;
}

}  // namespace shiftnd
