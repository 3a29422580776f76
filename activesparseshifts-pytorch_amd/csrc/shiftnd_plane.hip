// shiftnd_plane.hip -- the tuned per-(N,C)-plane kernels for gfx950 (MI355X).
//
// Design (HBM-bound gather; no MFMA):
//   * One workgroup owns one channel c and a group of batch entries n (optionally a band of rows of
//     a large plane).  The shift of a channel is the same for every n, so the workgroup evaluates
//     the padding index map ONCE per spatial dim into LDS (S_d + 1 int32 entries per dim):
//     map_d[p] = source index of coordinate p, or -1 for "fill".  All five padding modes and
//     arbitrarily large (multi-wrap) shifts cost the same in the streaming loop: the mode is only
//     visible in the prologue.
//   * The innermost (contiguous) dim is cut into 16-byte chunks; a thread owns one chunk column and
//     walks down the rows (rows of all planes of the group form one "super-row" sequence so small
//     planes still fill the workgroup).  Stores are aligned 16-byte row segments; loads are
//     16-byte global loads at element alignment (gfx950 global loads take any alignment), so an
//     arbitrary column shift costs no realignment work.  Only chunks that touch the padded region
//     (map not affine across the chunk) take the per-element gather path.
//   * Interpolation (active forward, backward) widens to fp32 (fp64 for fp64 tensors), evaluates
//     v1*(1-x)+v2*x without FMA contraction, rounds once on store.
//   * The weight gradient is accumulated per thread in fp64, reduced per workgroup (shuffle tree +
//     LDS) into a [group][C][3] partial buffer and summed by reduce_weight_grads: deterministic,
//     no atomics.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/):
//   forward   kernels/shifts_kernels.h:156-220, cuda/shifts_cuda.cu:202-266
//   backward  kernels/shifts_kernels.h:222-327, cuda/shifts_cuda.cu:270-345
//   quantized kernels/shifts_kernels.h:532-571, quantized/shifts_quantized.cpp:107-130
#include <algorithm>

#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"
#include "shiftnd_stage.hpp"

#ifndef SHIFTND_DMA_AUX_X
#define SHIFTND_DMA_AUX_X 0  // cache-policy bits of the LDS-DMA row loads (2 = nt); A/B builds override
#endif
#ifndef SHIFTND_DMA_AUX_G
#define SHIFTND_DMA_AUX_G 0
#endif

namespace shiftnd {
namespace {

struct PlaneParams {
    const void *x;      // forward: input; backward: saved input
    const void *go;     // backward: incoming gradient (forward output shape)
    void *out;          // forward: output; backward: grad_x
    const void *w;      // weights (float dtype or quantized int_repr)
    double *partials;   // backward: [groups*bands][C][3]
    int64_t wzp;
    uint64_t fill;
    int64_t x_plane;    // elements per (n,c) plane of x
    int64_t o_plane;    // elements per plane of out / grad_out
    int wkind, N, C, nd, pad;
    int S[3], O[3], L[3], wcol[3];
    int ppw, groups, bands, rows_per_band;
    int cpr, CW, RPS, CP;
    int rows;           // rows per plane of the iteration space
    int tile_bytes;         // LDS-staged backward: bytes of the row tile in front of the maps
    unsigned xppr;          // LDS-staged gather forward: 16-byte pieces per SOURCE row
    FastDiv d_xppr;
    int lds_affine;         // LDS-staged kernels: read affine chunks as consecutive dwords (tuning knob 5 = 1 turns it off)
    unsigned xcd_blocks;    // grid / 8 when the XCD-contiguous block remap is on (grid % 8 == 0), else 0
    FastDiv d_cpr;
    FastDiv d_rows;     // divide by rows_per_band
    FastDiv d_dim1;     // divide by the second outer dim of the iteration space
    FastDiv d_per[3];   // divide by the padding period of each dim of x (sizes S)
    FastDiv d_perO[3];  // ... of out / grad_out (sizes O)
    FastDiv d_C, d_groups;
    // fused average pool (kernel = stride = K, ceil mode; modules/shifts.py:81-89): pooled sizes P = ceil(O / K)
    int K[3], P[3];
    FastDiv d_k[3];
    FastDiv d_p2;
};

struct WorkItem {  // which planes / rows this workgroup owns
    int c, n0, nn, row0, nrows, pidx;
};
__device__ __forceinline__ WorkItem decode_block(const PlaneParams &p) {
    WorkItem wi;
    // XCD-contiguous ids (tuning knob 6): workgroups that share an XCD (blockIdx % 8) own adjacent planes
    const int bid = p.xcd_blocks ? static_cast<int>((blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3)) : static_cast<int>(blockIdx.x);
    const int rest = static_cast<int>(fdiv(static_cast<uint32_t>(bid), p.d_C));
    wi.c = bid - rest * p.C;
    const int band = static_cast<int>(fdiv(static_cast<uint32_t>(rest), p.d_groups));
    const int grp = rest - band * p.groups;
    wi.n0 = grp * p.ppw;
    wi.nn = min(p.ppw, p.N - wi.n0);
    wi.row0 = band * p.rows_per_band;
    wi.nrows = min(p.rows_per_band, p.rows - wi.row0);
    wi.pidx = band * p.groups + grp;
    return wi;
}

// =====================================================================================================
// Gather forward: SSL forward for every float dtype and the quantized forward (pure element copy).
// =====================================================================================================
template <int ESIZE, int V, int U>
__global__ __launch_bounds__(kThreads) void plane_gather_forward(const PlaneParams p) {
    using R = typename raw_t<ESIZE>::type;
    constexpr int E = V / ESIZE;
    extern __shared__ int maps[];
    const int *m0 = maps, *m1 = maps + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;

    const WorkItem wi = decode_block(p);
    int64_t sh[3];
    gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(wi.c) * p.nd, p.wcol, sh);
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d] = p.wcol[d] >= 0 ? sh[d] : 0;
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    __syncthreads();

    const R *__restrict__ x = static_cast<const R *>(p.x);
    R *__restrict__ out = static_cast<R *>(p.out);
    const R fill = static_cast<R>(p.fill);
    const int tr = threadIdx.x / p.CW, tc = threadIdx.x - tr * p.CW;
    if (tr >= p.RPS) return;
    const int SR = wi.nn * wi.nrows;
    const int S1 = p.S[1], S2 = p.S[2], O1 = p.O[1], O2 = p.O[2];

    for (int cp = 0; cp < p.CP; ++cp) {
        const int chunk = cp * p.CW + tc;
        if (chunk >= p.cpr) break;
        const int jo = chunk * E;
        int mm[E];
        bool contig = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mm[e] = m2[jo + p.L[2] + e];
            contig = contig && (mm[e] == mm[0] + e);
        }
        contig = contig && (mm[0] >= 0);

        for (int sr0 = tr; sr0 < SR; sr0 += p.RPS * U) {
            Chunk<R, E> v[U];
            R *dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sr = sr0 + u * p.RPS;
                dst[u] = nullptr;
                if (sr < SR) {
                    const int nl = fdiv(sr, p.d_rows);
                    const int r = wi.row0 + (sr - nl * wi.nrows);
                    const int a = fdiv(r, p.d_dim1);
                    const int b = r - a * O1;
                    const int ra = m0[a + p.L[0]], rb = m1[b + p.L[1]];
                    const int64_t plane = static_cast<int64_t>(wi.n0 + nl) * p.C + wi.c;
                    dst[u] = out + plane * p.o_plane + static_cast<int64_t>(r) * O2 + jo;
                    if (ra < 0 || rb < 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) v[u].e[e] = fill;
                    } else {
                        const R *row = x + plane * p.x_plane + static_cast<int64_t>(ra * S1 + rb) * S2;
                        if (contig) {
                            v[u] = load_chunk<R, E, true>(row + mm[0]);
                        } else {
#pragma unroll
                            for (int e = 0; e < E; ++e) v[u].e[e] = mm[e] >= 0 ? row[mm[e]] : fill;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u]) store_chunk<R, E>(dst[u], v[u]);
        }
    }
}

// =====================================================================================================
// Row loader for the interpolating kernels: E (+1) consecutive mapped elements of one source row.
// =====================================================================================================
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

// Outer-dim corner combos of a row: k bit r <-> +1 along real dim r (r < ND-1).
// Returns the row offset (elements, within the plane) or -1 when any outer map says "fill".
template <int ND>
__device__ __forceinline__ int combo_offset(int k, const int *m0, const int *m1, int pa, int pb, int stride0, int stride1) {
    if constexpr (ND == 1) {
        return 0;
    } else if constexpr (ND == 2) {
        const int rb = m1[pb + (k & 1)];
        return rb < 0 ? -1 : rb * stride1;
    } else {
        const int ra = m0[pa + (k & 1)];
        const int rb = m1[pb + ((k >> 1) & 1)];
        return (ra < 0 || rb < 0) ? -1 : ra * stride0 + rb * stride1;
    }
}

// =====================================================================================================
// Active forward: out = interp of the 2^ND corners around (coord - floor(w)).
// =====================================================================================================
// VB = bytes a thread moves per row step: 16, or -- rows that are not whole 16-byte pieces, round 5 -- 4 (two 16-bit elements, one
// fp32) / 8 (one fp64): the same kernel with narrower chunks (plane_ragged_forward / plane_ragged_backward below)
template <typename T, int ND, int VB = 16>
__global__ __launch_bounds__(kThreads) void plane_active_forward(const PlaneParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = VB / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
    extern __shared__ int maps[];
    const int *m0 = maps, *m1 = maps + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;

    const WorkItem wi = decode_block(p);
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(wi.c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            prep_shift_forward<CT>(wv[d], true, sh[d], dw[p.wcol[d]]);
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    __syncthreads();

    const S *__restrict__ x = static_cast<const S *>(p.x);
    S *__restrict__ out = static_cast<S *>(p.out);
    const int tr = threadIdx.x / p.CW, tc = threadIdx.x - tr * p.CW;
    if (tr >= p.RPS) return;
    const int SR = wi.nn * wi.nrows;
    const int S1 = p.S[1], S2 = p.S[2], O1 = p.O[1], O2 = p.O[2];

    for (int cp = 0; cp < p.CP; ++cp) {
        const int chunk = cp * p.CW + tc;
        if (chunk >= p.cpr) break;
        const int jo = chunk * E;
        int mm[E + 1];
        bool contig = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) {
            mm[e] = m2[jo + p.L[2] + e];
            if (e < E) contig = contig && (mm[e] == mm[0] + e);
        }
        contig = contig && (mm[0] >= 0);

        for (int sr = tr; sr < SR; sr += p.RPS) {
            const int nl = fdiv(sr, p.d_rows);
            const int r = wi.row0 + (sr - nl * wi.nrows);
            const int a = fdiv(r, p.d_dim1);
            const int b = r - a * O1;
            const int64_t plane = static_cast<int64_t>(wi.n0 + nl) * p.C + wi.c;
            const S *xp = x + plane * p.x_plane;
            CT vals[NC][E + 1];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int off = combo_offset<ND>(k, m0, m1, a + p.L[0], b + p.L[1], S1 * S2, S2);
                load_row<T, E, E + 1>(xp + (off < 0 ? 0 : off), off >= 0, contig, mm, vals[k]);
            }
            Chunk<S, E> res;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[1 << ND];
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) v[q] = vals[q & (NC - 1)][e + (q >> (ND - 1))];
                res.e[e] = narrow<T>(interp_t<T, ND>(v, dw));
            }
            store_chunk<S, E>(out + plane * p.o_plane + static_cast<int64_t>(r) * O2 + jo, res);
        }
    }
}


// =====================================================================================================
// Pooled forward: the shift followed by the average pool the reference's modules attach when they emulate a
// strided depthwise conv (modules/shifts.py:81-89, 150-153: avg_pool{N}d(kernel = stride = K, ceil_mode=True))
// in one pass: the shift output (K0*K1*K2 times larger than the result) never goes to HBM.
// One thread per pooled element; the window is summed in ATen's order (row-major, starting from 0) in the
// compute type and divided by the number of window elements inside the shift output, so fp32 / fp64 results
// are bit-identical to shift + avg_pool.  16-bit interpolated values are rounded to the storage type first,
// like the unfused sequence does when it stores the shift output.
// =====================================================================================================
// SMALLK: every window size is 1 or 2 (the strided-conv emulation with stride 2).  Branch-free form: the source
// samples a pooled element needs -- a 2 x 2 (x 2) block, one more per real dim when interpolating, since adjacent
// window positions share corners (9 loads instead of 16 in 2-D, 27 instead of 64 in 3-D) -- are loaded as
// independent, predicated loads, then combined from registers.
template <typename T, int ND, bool ACTIVE, bool SMALLK>
__global__ __launch_bounds__(kThreads) void plane_pool_forward(const PlaneParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    extern __shared__ int maps[];
    const int *m0 = maps, *m1 = maps + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;

    const WorkItem wi = decode_block(p);  // rows = pooled rows P0 * P1
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(wi.c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            prep_shift_forward<CT>(wv[d], ACTIVE, sh[d], dw[p.wcol[d]]);
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    __syncthreads();

    const S *__restrict__ x = static_cast<const S *>(p.x);
    S *__restrict__ out = static_cast<S *>(p.out);
    const int S1 = p.S[1], S2 = p.S[2], P1 = p.P[1], P2 = p.P[2];
    const int total = wi.nn * wi.nrows * P2;
    for (int t = threadIdx.x; t < total; t += kThreads) {
        const int sr = fdiv(t, p.d_p2);
        const int p2 = t - sr * P2;
        const int nl = fdiv(sr, p.d_rows);
        const int r = wi.row0 + (sr - nl * wi.nrows);
        const int p0 = fdiv(r, p.d_dim1);
        const int p1 = r - p0 * P1;
        const int64_t plane = static_cast<int64_t>(wi.n0 + nl) * p.C + wi.c;
        const S *xp = x + plane * p.x_plane;
        const int n0 = min(p.K[0], p.O[0] - p0 * p.K[0]), n1 = min(p.K[1], p.O[1] - p1 * p.K[1]);
        const int n2 = min(p.K[2], p.O[2] - p2 * p.K[2]);
        CT acc = CT(0);
        if constexpr (SMALLK) {
            // window extent M and sample extent A per normalised dim (dims in front of the real ones have size 1)
            constexpr int M0 = ND == 3 ? 2 : 1, M1 = ND >= 2 ? 2 : 1, M2 = 2;
            constexpr int A0 = M0 + (ACTIVE && ND == 3 ? 1 : 0), A1 = M1 + (ACTIVE && ND >= 2 ? 1 : 0), A2 = M2 + (ACTIVE ? 1 : 0);
            int r0[A0], r1[A1], r2[A2];
            // (a sample index beyond the last one the window uses is clamped to the map's last entry and never used)
#pragma unroll
            for (int a = 0; a < A0; ++a) r0[a] = m0[min(p0 * p.K[0] + p.L[0] + a, p.S[0])];
#pragma unroll
            for (int a = 0; a < A1; ++a) r1[a] = m1[min(p1 * p.K[1] + p.L[1] + a, S1)];
#pragma unroll
            for (int a = 0; a < A2; ++a) r2[a] = m2[min(p2 * p.K[2] + p.L[2] + a, S2)];
            CT sv[A0][A1][A2];
#pragma unroll
            for (int a = 0; a < A0; ++a)
#pragma unroll
                for (int b = 0; b < A1; ++b) {
                    const bool rowok = r0[a] >= 0 && r1[b] >= 0;
                    const int rowoff = rowok ? (r0[a] * S1 + r1[b]) * S2 : 0;
#pragma unroll
                    for (int c = 0; c < A2; ++c) {
                        const bool ok = rowok && r2[c] >= 0;
                        const CT v = widen<T>(xp[rowoff + (r2[c] >= 0 ? r2[c] : 0)]);
                        sv[a][b][c] = ok ? v : CT(0);
                    }
                }
#pragma unroll
            for (int u0 = 0; u0 < M0; ++u0)
#pragma unroll
                for (int u1 = 0; u1 < M1; ++u1)
#pragma unroll
                    for (int u2 = 0; u2 < M2; ++u2) {
                        CT val;
                        if constexpr (ACTIVE) {
                            CT v[1 << ND];
#pragma unroll
                            for (int q = 0; q < (1 << ND); ++q) {
                                // bit r of q <-> +1 along real dim r = normalised dim r + 3 - ND
                                const int b0 = ND == 3 ? (q & 1) : 0;
                                const int b1 = ND == 3 ? ((q >> 1) & 1) : (ND == 2 ? (q & 1) : 0);
                                const int b2 = (q >> (ND - 1)) & 1;
                                v[q] = sv[u0 + b0][u1 + b1][u2 + b2];
                            }
                            val = widen<T>(narrow<T>(interp_t<T, ND>(v, dw)));
                        } else {
                            val = sv[u0][u1][u2];
                        }
                        // (acc is never -0.0, so adding +0.0 for a position outside a ragged last window changes nothing)
                        acc = acc + ((u0 < n0 && u1 < n1 && u2 < n2) ? val : CT(0));
                    }
        } else {
            for (int u0 = 0; u0 < n0; ++u0) {
                const int i0 = p0 * p.K[0] + u0 + p.L[0];
                for (int u1 = 0; u1 < n1; ++u1) {
                    const int i1 = p1 * p.K[1] + u1 + p.L[1];
                    for (int u2 = 0; u2 < n2; ++u2) {
                        const int i2 = p2 * p.K[2] + u2 + p.L[2];
                        if constexpr (ACTIVE) {
                            CT v[1 << ND];
#pragma unroll
                            for (int q = 0; q < (1 << ND); ++q) {
                                const int b0 = ND == 3 ? (q & 1) : 0;
                                const int b1 = ND == 3 ? ((q >> 1) & 1) : (ND == 2 ? (q & 1) : 0);
                                const int b2 = (q >> (ND - 1)) & 1;
                                const int ra = m0[i0 + b0], rb = m1[i1 + b1], rc = m2[i2 + b2];
                                v[q] = (ra >= 0 && rb >= 0 && rc >= 0) ? widen<T>(xp[(ra * S1 + rb) * S2 + rc]) : CT(0);
                            }
                            acc = acc + widen<T>(narrow<T>(interp_t<T, ND>(v, dw)));
                        } else {
                            const int ra = m0[i0], rb = m1[i1], rc = m2[i2];
                            acc = acc + ((ra >= 0 && rb >= 0 && rc >= 0) ? widen<T>(xp[(ra * S1 + rb) * S2 + rc]) : CT(0));
                        }
                    }
                }
            }
        }
        out[plane * p.o_plane + static_cast<int64_t>(r) * P2 + p2] = narrow<T>(div_count<CT>(acc, n0 * n1 * n2));
    }
}

// =====================================================================================================
// Backward: grad_x (gather of grad_out, SSL; or interpolation of grad_out, active) and the weight
// gradient partials.  Iteration space = input coordinates.
// LDS maps: x maps (3 dims, S_d + 1 entries) followed by grad_out maps (3 dims, O_d + 1 entries).
// =====================================================================================================
// POOL: grad_out is the gradient of the POOLED output [P0][P1][P2]; the gradient of the shift output it stands for
// is g(o) = round_S(grad_pooled[o / K] / count(o / K)) (ATen's avg_pool backward), evaluated on the fly.
template <typename T, int ND>
__device__ __forceinline__ bool combo_rows(int k, const int *m0, const int *m1, int pa, int pb, int &ra, int &rb) {
    if constexpr (ND == 1) {
        ra = 0;
        rb = 0;
        return true;
    } else if constexpr (ND == 2) {
        ra = 0;
        rb = m1[pb + (k & 1)];
        return rb >= 0;
    } else {
        ra = m0[pa + (k & 1)];
        rb = m1[pb + ((k >> 1) & 1)];
        return ra >= 0 && rb >= 0;
    }
}
struct PoolRow {  // one row of the unpooled gradient: where it lives in the pooled plane
    int off;  // element offset of the pooled row
    int cnt;  // window elements along the two outer dims
};
__device__ __forceinline__ PoolRow pool_row(const PlaneParams &p, int a, int b) {
    const int pa = fdiv(a, p.d_k[0]), pb = fdiv(b, p.d_k[1]);
    PoolRow r;
    r.off = (pa * p.P[1] + pb) * p.P[2];
    r.cnt = min(p.K[0], p.O[0] - pa * p.K[0]) * min(p.K[1], p.O[1] - pb * p.K[1]);
    return r;
}
template <typename T>
__device__ __forceinline__ typename T::C pool_grad(const typename T::S *gp, const PoolRow &r, int pc, int cc) {
    using CT = typename T::C;
    return widen<T>(narrow<T>(div_count<CT>(widen<T>(gp[r.off + pc]), r.cnt * cc)));
}

template <typename T, int ND, bool ACTIVE, bool POOL = false, int VB = 16>
__global__ __launch_bounds__(kThreads) void plane_backward(const PlaneParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = VB / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
    extern __shared__ int maps[];
    __shared__ double scratch[kThreads / 64];
    const int *m0 = maps, *m1 = m0 + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;
    int *gmaps = maps + p.S[0] + p.S[1] + p.S[2] + 3;
    const int *g0 = gmaps, *g1 = g0 + p.O[0] + 1, *g2 = g1 + p.O[1] + 1;

    const WorkItem wi = decode_block(p);
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(wi.c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            prep_shift_backward<CT>(wv[d], ACTIVE, sh[d], dw[p.wcol[d]]);
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    // grad_x source: SSL reads grad_out at o + shift, active at o - shift (shifts_kernels.h:287-293)
    build_maps(gmaps, p.O, sh, ACTIVE ? -1 : +1, p.pad, p.d_perO);
    __syncthreads();

    const S *__restrict__ x = static_cast<const S *>(p.x);
    const S *__restrict__ go = static_cast<const S *>(p.go);
    S *__restrict__ gx = static_cast<S *>(p.out);
    const int tr = threadIdx.x / p.CW, tc = threadIdx.x - tr * p.CW;
    const int SR = wi.nn * wi.nrows;
    const int S1 = p.S[1], S2 = p.S[2], O1 = p.O[1], O2 = p.O[2];
    double acc[3] = {0.0, 0.0, 0.0};

    if (tr < p.RPS) {
        for (int cp = 0; cp < p.CP; ++cp) {
            const int chunk = cp * p.CW + tc;
            if (chunk >= p.cpr) break;
            const int ji = chunk * E;   // input inner coordinate of element 0
            const int oj = ji - p.L[2];  // grad_out inner coordinate of element 0 (may be outside)
            // per-chunk column state -----------------------------------------------------------------
            int xm[E + 1], gm[E + 1];
            int pcd[POOL ? E : 1], ccd[POOL ? E : 1], pcm[POOL ? E + 1 : 1], ccm[POOL ? E + 1 : 1];  // pooled column / window width
            unsigned inmask = 0;
            bool xcontig = true, gcontig = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                xm[e] = m2[ji + e];
                if (e < E) xcontig = xcontig && (xm[e] == xm[0] + e);
                const int o = oj + e;
                const bool in = (o >= 0) && (o < O2);
                if (e < E && in) inmask |= 1u << e;
                // grad_out map; clamp the LDS index, entries of outside elements are never used
                const int oc = o < 0 ? 0 : (o > O2 ? O2 : o);
                gm[e] = g2[oc];
                if (e < E) gcontig = gcontig && (gm[e] == gm[0] + e);
                if constexpr (POOL) {
                    if (e < E) {
                        pcd[e] = fdiv(in ? o : 0, p.d_k[2]);
                        ccd[e] = min(p.K[2], O2 - pcd[e] * p.K[2]);
                    }
                    pcm[e] = fdiv(gm[e] < 0 ? 0 : gm[e], p.d_k[2]);
                    ccm[e] = min(p.K[2], O2 - pcm[e] * p.K[2]);
                }
            }
            const bool allin = inmask == ((1u << E) - 1u);
            xcontig = xcontig && (xm[0] >= 0);
            gcontig = gcontig && allin && (gm[0] >= 0);

            for (int sr = tr; sr < SR; sr += p.RPS) {
                const int nl = fdiv(sr, p.d_rows);
                const int r = wi.row0 + (sr - nl * wi.nrows);
                const int a = fdiv(r, p.d_dim1);
                const int b = r - a * S1;
                const int oa = a - p.L[0], ob = b - p.L[1];
                const int64_t plane = static_cast<int64_t>(wi.n0 + nl) * p.C + wi.c;
                S *dst = gx + plane * p.x_plane + static_cast<int64_t>(r) * S2 + ji;
                Chunk<S, E> res;
                const bool rowin = (oa >= 0) && (oa < p.O[0]) && (ob >= 0) && (ob < O1);
                if (!rowin || inmask == 0) {  // outside the border window: grad_x = 0, no weight-grad term
#pragma unroll
                    for (int e = 0; e < E; ++e) res.e[e] = narrow<T>(CT(0));
                    store_chunk<S, E>(dst, res);
                    continue;
                }
                const S *xp = x + plane * p.x_plane;
                const S *gp = go + plane * p.o_plane;
                // incoming gradient at this position ----------------------------------------------------
                CT gval[E];
                if constexpr (POOL) {
                    const PoolRow pr = pool_row(p, oa, ob);
#pragma unroll
                    for (int e = 0; e < E; ++e) gval[e] = ((inmask >> e) & 1u) ? pool_grad<T>(gp, pr, pcd[e], ccd[e]) : CT(0);
                } else {
                    const S *grow = gp + static_cast<int64_t>(oa * O1 + ob) * O2;
                    if (allin) {
                        const Chunk<S, E> c = load_chunk<S, E>(grow + oj);
#pragma unroll
                        for (int e = 0; e < E; ++e) gval[e] = widen<T>(c.e[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) gval[e] = ((inmask >> e) & 1u) ? widen<T>(grow[oj + e]) : CT(0);
                    }
                }
                // weight gradient: corners of x around (coord - shift) --------------------------------------
                {
                    CT vals[NC][E + 1];
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const int off = combo_offset<ND>(k, m0, m1, a, b, S1 * S2, S2);
                        load_row<T, E, E + 1>(xp + (off < 0 ? 0 : off), off >= 0, xcontig, xm, vals[k]);
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        CT v[1 << ND], wg[3];
#pragma unroll
                        for (int q = 0; q < (1 << ND); ++q) v[q] = vals[q & (NC - 1)][e + (q >> (ND - 1))];
                        weight_grads_nd<ND, CT>(v, dw, wg);
                        if ((inmask >> e) & 1u) {
#pragma unroll
                            for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval[e] * wg[s]);
                        }
                    }
                }
                // input gradient ------------------------------------------------------------------------
                if constexpr (ACTIVE) {
                    CT vals[NC][E + 1];
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        if constexpr (POOL) {
                            int ra, rb;
                            const bool ok = combo_rows<T, ND>(k, g0, g1, oa, ob, ra, rb);
                            const PoolRow pr = pool_row(p, ok ? ra : 0, ok ? rb : 0);
#pragma unroll
                            for (int e = 0; e <= E; ++e) vals[k][e] = (ok && gm[e] >= 0) ? pool_grad<T>(gp, pr, pcm[e], ccm[e]) : CT(0);
                        } else {
                            const int off = combo_offset<ND>(k, g0, g1, oa, ob, O1 * O2, O2);
                            load_row<T, E, E + 1>(gp + (off < 0 ? 0 : off), off >= 0, gcontig, gm, vals[k]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        CT v[1 << ND];
#pragma unroll
                        for (int q = 0; q < (1 << ND); ++q) v[q] = vals[q & (NC - 1)][e + (q >> (ND - 1))];
                        const CT r1 = interp_t<T, ND>(v, dw);
                        res.e[e] = narrow<T>(((inmask >> e) & 1u) ? r1 : CT(0));
                    }
                } else {
                    const int ra = g0[oa], rb = g1[ob];
                    if (ra < 0 || rb < 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) res.e[e] = narrow<T>(CT(0));
                    } else if constexpr (POOL) {
                        const PoolRow pr = pool_row(p, ra, rb);
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            res.e[e] = narrow<T>((((inmask >> e) & 1u) && gm[e] >= 0) ? pool_grad<T>(gp, pr, pcm[e], ccm[e]) : CT(0));
                    } else {
                        const S *srow = gp + static_cast<int64_t>(ra * O1 + rb) * O2;
                        if (gcontig) {
                            res = load_chunk<S, E>(srow + gm[0]);
                        } else {
#pragma unroll
                            for (int e = 0; e < E; ++e)
                                res.e[e] = (((inmask >> e) & 1u) && gm[e] >= 0) ? srow[gm[e]] : narrow<T>(CT(0));
                        }
                    }
                }
                store_chunk<S, E>(dst, res);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double t = block_sum(acc[s], scratch);
        if (threadIdx.x == 0) p.partials[(static_cast<size_t>(wi.pidx) * p.C + wi.c) * 3 + s] = t;
    }
}

// =====================================================================================================
// LDS-staged kernels (2-D and 3-D, no crop): every source row a step needs is brought into LDS once with
// aligned 16-byte LDS-DMA loads (global_load_lds_dwordx4: no VGPRs, whole lines), and all shifted / corner /
// padded accesses become ds_reads at element granularity.  Compared with the direct-load kernels this removes
// the element-aligned global loads (every line requested twice), the per-element edge path (one ds_read code
// path serves interior and padded chunks) and the repeated global reads of rows that neighbouring rows share.
// A step = up to R consecutive rows (a, b0 .. b0+Rn-1) of one plane (fixed a).  Slots of its tile:
//   X   NA x (R+1) rows of x:  (map0[a + ha], map1[b0 + k])            NA = 2 for 3-D (corner planes), else 1
//   G   R rows of grad_out at the rows themselves (backward only)
//   GS  backward, active: NA x (R+1) rows of grad_out through the grad maps; SSL: R shifted rows
// A slot whose map says "fill" is not loaded and never read.
// =====================================================================================================
// SCAT (2-D sparse-shift backward): no GS slots.  The sparse shift's input gradient is a row permutation of grad_out
// (plus fill / folded rows at the two ends of a plane), so the G rows a step stages for the weight gradient are also
// written out as the grad_x rows that read them (grad_x row b - shift <- grad_out row b): every grad_out row crosses
// the L2 -> LDS path once instead of twice (13 -> 9 staged rows per 4-row step).
template <int ND, bool ACTIVE, bool BACKWARD, bool SCAT = false> struct LdsTileShape {
    static constexpr int NA = ND == 3 ? 2 : 1;
    __host__ __device__ static int nx(int R) { return NA * (R + 1); }
    __host__ __device__ static int ng(int R) { return BACKWARD ? R : 0; }
    __host__ __device__ static int ngs(int R) { return (BACKWARD && !SCAT) ? (ACTIVE ? NA * (R + 1) : R) : 0; }
    __host__ __device__ static int slots(int R) { return nx(R) + ng(R) + ngs(R); }
};

// Shared by the LDS-staged kernels: slot table + LDS-DMA of one step.
template <typename T, int ND, bool ACTIVE, bool BACKWARD, bool POOL = false, bool SCAT = false>
struct LdsStager {
    using S = typename T::S;
    using Shape = LdsTileShape<ND, ACTIVE, BACKWARD, SCAT>;
    static constexpr int E = 16 / sizeof(S);
    static constexpr int NA = Shape::NA;

    // slot table: element offset of each staged row inside its plane, or -1
    // POOL: a G / GS entry is the element offset of the POOLED row the slot's (unpooled) gradient row expands from,
    // and slot_src[k + aux] the number of window elements along the two outer dims (see expand_pooled)
    __device__ static void make_slots(const PlaneParams &p, int R, int a, int b0, int Rn, const int *m0, const int *m1,
                                      const int *g0, const int *g1, int *slot_src, int aux = 0) {
        const int NX = Shape::nx(R), NG = Shape::ng(R), NS = Shape::slots(R);
        const int k = threadIdx.x;
        if (k >= NS) return;
        int src = -1;
        auto grad_row = [&](int ra, int rb) {
            if constexpr (POOL) {
                const PoolRow pr = pool_row(p, ra, rb);
                slot_src[k + aux] = pr.cnt;
                return pr.off;
            } else {
                return (ra * p.S[1] + rb) * p.S[2];
            }
        };
        if (k < NX) {
            const int ha = k / (R + 1), kk = k - ha * (R + 1);
            if (kk <= Rn) {
                const int ra = m0[a + ha], rb = m1[b0 + kk];
                if (ra >= 0 && rb >= 0) src = (ra * p.S[1] + rb) * p.S[2];
            }
        } else if (k < NX + NG) {
            const int kk = k - NX;
            if (kk < Rn) src = grad_row(a, b0 + kk);
        } else if (ACTIVE) {
            const int k2 = k - NX - NG;
            const int ha = k2 / (R + 1), kk = k2 - ha * (R + 1);
            if (kk <= Rn) {
                const int ra = g0[a + ha], rb = g1[b0 + kk];
                if (ra >= 0 && rb >= 0) src = grad_row(ra, rb);
            }
        } else {
            const int kk = k - NX - NG;
            if (kk < Rn) {
                const int ra = g0[a], rb = g1[b0 + kk];
                if (ra >= 0 && rb >= 0) src = grad_row(ra, rb);
            }
        }
        slot_src[k] = src;
    }

    // POOL: a G / GS slot holds one row of the UNPOOLED gradient g(o) = round_S(grad_pooled[o / K] / count(o / K)),
    // expanded here from the pooled row (piece j = columns [j*E, j*E + E)), so the compute phase reads the tile
    // exactly as it reads staged grad_out rows.  Windows of 2 along the row (the stride-2 emulation) take one load
    // of E/2 pooled values per piece and, when the count is a power of two, one multiply each.
    __device__ static void expand_pooled(const PlaneParams &p, const S *prow, int row_cnt, int j, char *dst) {
        using CT = typename T::C;
        Chunk<S, E> out;
        if (p.K[2] == 2 && E % 2 == 0) {  // (rows are whole 16-byte pieces, so O2 is even: every window has 2 columns)
            const Chunk<S, (E >= 2 ? E / 2 : 1)> pv = load_chunk<S, (E >= 2 ? E / 2 : 1)>(prow + j * (E / 2));
#pragma unroll
            for (int h = 0; h < E / 2; ++h) {
                const S q = narrow<T>(div_count<CT>(widen<T>(pv.e[h]), row_cnt * 2));
                out.e[2 * h] = q;
                out.e[2 * h + 1] = q;
            }
        } else {
            const int c0 = j * E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pc = fdiv(c0 + e, p.d_k[2]);
                const int cc = min(p.K[2], p.O[2] - pc * p.K[2]);
                out.e[e] = narrow<T>(div_count<CT>(widen<T>(prow[pc]), row_cnt * cc));
            }
        }
        __builtin_memcpy(__builtin_assume_aligned(dst, 16), out.e, 16);
    }

    // LDS-DMA: piece q = 16 bytes; lanes of a wave take consecutive pieces (the LDS destination is linear)
    // Which pieces a thread moves never changes from step to step: piece k of a thread is q = k * kThreads + tid.
    // Its (slot, column piece) pair is decoded once, packed as slot * 256 + j (or -1), and the per-step work is the
    // slot-table read, one address and the DMA.  For steps of at most kRegPieces pieces per thread (cpr <= 256).
    // (3 keeps the 2-D SSL fp32 kernel at 64 VGPRs; the 3-D kernels stage up to 49 short rows per step)
    static constexpr int kRegPieces = ND == 3 ? 6 : ((ACTIVE || POOL || sizeof(S) == 2) ? 4 : (SCAT ? 2 : 3));
    __host__ __device__ static bool pieces_fit(int cpr, int R) {
        return Shape::slots(R) * cpr <= kRegPieces * kThreads && cpr <= 256;
    }
    __device__ static void decode_pieces(const PlaneParams &p, int R, int (&pk)[kRegPieces]) {
        const int pieces = Shape::slots(R) * static_cast<int>(p.cpr);
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            const int q = k * kThreads + static_cast<int>(threadIdx.x);
            const int slot = fdiv(q, p.d_cpr);
            pk[k] = q < pieces ? slot * 256 + (q - slot * static_cast<int>(p.cpr)) : -1;
        }
    }
    __device__ static void issue_dma_decoded(int NX, const S *xp, const S *gp, const int *slot_src, char *tile,
                                             const int (&pk)[kRegPieces]) {
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            if (pk[k] >= 0) {
                const int slot = pk[k] >> 8;
                const int src = slot_src[slot];
                if (src >= 0) {
                    // uniform base + 32-bit lane offset (planes are < 2^30 elements): the SGPR-base address form, one
                    // instruction per array instead of a per-lane 64-bit pointer select
                    const uint32_t off = static_cast<uint32_t>(src + (pk[k] & 255) * E) * static_cast<uint32_t>(sizeof(S));
                    char *dst_wave = tile + (k * kThreads + wave * 64) * 16;  // wave-uniform; hardware adds lane * 16
                    if (slot < NX)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, SHIFTND_DMA_AUX_X);
                    else
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(gp) + off),
                                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, SHIFTND_DMA_AUX_G);
                }
            }
        }
    }

    // POOL with windows of 2 along the row, pre-decoded pieces: x pieces go by LDS-DMA; the pooled values of ALL of the
    // thread's gradient pieces are loaded first and expanded afterwards, so their latencies overlap each other and
    // the DMA instead of being paid piece by piece.
    __device__ static void issue_dma_decoded_pool(int NX, const S *xp, const S *gp, const int *slot_src, char *tile,
                                                  const int (&pk)[kRegPieces], int aux) {
        using CT = typename T::C;
        constexpr int H = E >= 2 ? E / 2 : 1;
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
        Chunk<S, H> pv[kRegPieces];
        int cnt[kRegPieces];
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            cnt[k] = 0;
            if (pk[k] >= 0) {
                const int slot = pk[k] >> 8;
                const int src = slot_src[slot];
                if (src >= 0) {
                    if (slot < NX) {
                        const S *g = xp + src + (pk[k] & 255) * E;
                        char *dst_wave = tile + (k * kThreads + wave * 64) * 16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
                    } else {
                        pv[k] = load_chunk<S, H>(gp + src + (pk[k] & 255) * H);
                        cnt[k] = slot_src[slot + aux] * 2;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            if (cnt[k] > 0) {
                Chunk<S, E> out;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const S q = narrow<T>(div_count<CT>(widen<T>(pv[k].e[h]), cnt[k]));
                    out.e[2 * h] = q;
                    if (E >= 2) out.e[2 * h + 1] = q;
                }
                __builtin_memcpy(__builtin_assume_aligned(tile + (k * kThreads + static_cast<int>(threadIdx.x)) * 16, 16), out.e, 16);
            }
        }
    }

    // The same in three parts, for kernels that keep few workgroups resident (16-bit data): the pooled values of step
    // s + 1 are requested right after the barrier of step s, ride out their latency behind step s's compute phase, and
    // are expanded at the top of step s + 1.
    struct PoolRegs {
        Chunk<S, (E >= 2 ? E / 2 : 1)> pv[kRegPieces];
        int cnt[kRegPieces];
    };
    __device__ static void pool_load(int NX, const S *gp, const int *slot_src, const int (&pk)[kRegPieces], int aux, PoolRegs &r) {
        constexpr int H = E >= 2 ? E / 2 : 1;
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            r.cnt[k] = 0;
            if (pk[k] >= 0) {
                const int slot = pk[k] >> 8;
                const int src = slot_src[slot];
                if (src >= 0 && slot >= NX) {
                    r.pv[k] = load_chunk<S, H>(gp + src + (pk[k] & 255) * H);
                    r.cnt[k] = slot_src[slot + aux] * 2;
                }
            }
        }
    }
    __device__ static void pool_expand(const PoolRegs &r, char *tile) {
        using CT = typename T::C;
        constexpr int H = E >= 2 ? E / 2 : 1;
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            if (r.cnt[k] > 0) {
                Chunk<S, E> out;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const S q = narrow<T>(div_count<CT>(widen<T>(r.pv[k].e[h]), r.cnt[k]));
                    out.e[2 * h] = q;
                    if (E >= 2) out.e[2 * h + 1] = q;
                }
                __builtin_memcpy(__builtin_assume_aligned(tile + (k * kThreads + static_cast<int>(threadIdx.x)) * 16, 16), out.e, 16);
            }
        }
    }
    __device__ static void dma_x_decoded(int NX, const S *xp, const int *slot_src, char *tile, const int (&pk)[kRegPieces]) {
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
#pragma unroll
        for (int k = 0; k < kRegPieces; ++k) {
            if (pk[k] >= 0 && (pk[k] >> 8) < NX) {
                const int src = slot_src[pk[k] >> 8];
                if (src >= 0) {
                    const S *g = xp + src + (pk[k] & 255) * E;
                    char *dst_wave = tile + (k * kThreads + wave * 64) * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
                }
            }
        }
    }

    __device__ static void issue_dma(const PlaneParams &p, int R, const S *xp, const S *gp, const int *slot_src, char *tile,
                                     int aux = 0) {
        const int NX = Shape::nx(R);
        const int pieces = Shape::slots(R) * static_cast<int>(p.cpr);
        for (int q0 = 0; q0 < pieces; q0 += kThreads) {
            const int q = q0 + threadIdx.x;
            if (q < pieces) {
                const int slot = fdiv(q, p.d_cpr);
                const int j = q - slot * static_cast<int>(p.cpr);
                const int src = slot_src[slot];
                if (POOL && slot >= NX) {
                    if (src >= 0) expand_pooled(p, gp + src, slot_src[slot + aux], j, tile + q * 16);
                } else if (src >= 0) {
                    const S *g = (slot < NX ? xp : gp) + src + j * E;
                    char *dst_wave = tile + (q0 + (threadIdx.x & ~63)) * 16;  // wave-uniform; hardware adds lane * 16
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
                }
            }
        }
    }
};

// corner values of one chunk from the staged rows: vals[k][e], k = outer corner combo (bit r <-> +1 along real
// dim r < ND-1), e = 0..E (E + 1 columns through the column map)
template <typename T, int ND>
__device__ __forceinline__ void lds_corners(const char *tile, int RB, int R, int slot0, const int *ss, int tr,
                                            const ColState<16 / sizeof(typename T::S)> &cs,
                                            typename T::C (&vals)[1 << (ND - 1)][16 / sizeof(typename T::S) + 1]) {
    using S = typename T::S;
    constexpr int E = 16 / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int ha = ND == 3 ? (k & 1) : 0;
        const int hb = ND == 3 ? ((k >> 1) & 1) : (k & 1);
        const int slot = slot0 + ha * (R + 1) + tr + hb;
        S raw[E + 1];
        lds_read_row<S, E>(tile + slot * RB, ss[slot] >= 0, cs, raw);
#pragma unroll
        for (int e = 0; e <= E; ++e) vals[k][e] = widen<T>(raw[e]);
    }
}

// DEC: the step's DMA pieces per thread fit the pre-decoded form (LdsStager::pieces_fit, checked by the host)
template <typename T, int ND, bool ACTIVE, int TILES, bool POOL = false, bool DEC = false, bool SCAT = false>
__global__ __launch_bounds__(kThreads) void plane_backward_lds(const PlaneParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    static_assert(!SCAT || (ND == 2 && !ACTIVE && !POOL), "the scatter form serves the 2-D sparse-shift backward");
    using Stager = LdsStager<T, ND, ACTIVE, true, POOL, SCAT>;
    using Shape = LdsTileShape<ND, ACTIVE, true, SCAT>;
    constexpr int E = 16 / sizeof(S);
    constexpr int NC = 1 << (ND - 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double scratch[kThreads / 64];
    const int R = p.RPS;
    const int NX = Shape::nx(R), NG = Shape::ng(R), NS = Shape::slots(R);
    const int RB = p.S[2] * static_cast<int>(sizeof(S));  // row bytes, a multiple of 16
    char *tile0 = smem + 64;  // 64-byte pad: see lds_read_row
    int *maps = reinterpret_cast<int *>(smem + 64 + TILES * p.tile_bytes);  // TILES == 2: alternate tiles, one barrier per step
    const int *m0 = maps, *m1 = maps + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;
    int *gmaps = maps + p.S[0] + p.S[1] + p.S[2] + 3;
    const int *g0 = gmaps, *g1 = gmaps + p.O[0] + 1, *g2 = g1 + p.O[1] + 1;
    int *slot_src = gmaps + p.O[0] + p.O[1] + p.O[2] + 3;  // two tables of NS entries

    const WorkItem wi = decode_block(p);
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    CT wv[3];
    load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(wi.c) * p.nd, p.wcol, wv);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            prep_shift_backward<CT>(wv[d], ACTIVE, sh[d], dw[p.wcol[d]]);
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    build_maps(gmaps, p.O, sh, ACTIVE ? -1 : +1, p.pad, p.d_perO);
    __syncthreads();

    const int S1 = p.S[1], S2 = p.S[2];
    const int tr = threadIdx.x / p.CW, tc = threadIdx.x - tr * p.CW;
    const bool worker = tr < R;
    const int ji = tc * E;
    const ColState<E> xm = make_colstate<E>(m2, ji, worker, p.lds_affine != 0);
    const ColState<E> gm = make_colstate<E>(g2, ji, worker, p.lds_affine != 0);  // no crop: grad_out coordinates == input coordinates
    constexpr int NDIFF = WDiff<ND>::N;
    double dsum[NDIFF];  // sums of g * corner difference (see corner_diffs)
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) dsum[i] = 0.0;
    // SCAT: the row shift as the grad_x <- grad_out row map sees it (size-1 dims ignore the shift; anything beyond the
    // dim leaves no row that maps by the plain shift)
    const int scat_shift = (!SCAT || S1 == 1) ? 0 : static_cast<int>(sh[1] > S1 ? S1 : (sh[1] < -S1 ? -S1 : sh[1]));
    const int row_end = wi.row0 + wi.nrows;

    // steps never cross an `a` boundary; the slot table of step s+1 is written while step s is computed
    auto step_len = [&](int r0) {
        const int b0 = r0 - fdiv(r0, p.d_dim1) * S1;
        return min(R, min(S1 - b0, row_end - r0));
    };
    // slot tables rotate over NT buffers: with two tiles there is no barrier after the compute phase, so the table
    // of step s must survive until every wave has passed the barrier of step s+1 (three tables)
    constexpr int NT = TILES == 2 ? 3 : 2;
    const int aux = NT * NS;  // POOL: a second set of tables behind the slot tables
    int pk[Stager::kRegPieces];
    if constexpr (DEC) Stager::decode_pieces(p, R, pk);
    constexpr bool kPrefetchPool = DEC && POOL && sizeof(S) == 2;  // (fp32: 1.67 -> 2.09 ms with it)
    typename Stager::PoolRegs pregs;
    int nl = 0, r0 = wi.row0, buf = 0, tb = 0;
    {
        const int a = fdiv(r0, p.d_dim1);
        Stager::make_slots(p, R, a, r0 - a * S1, step_len(r0), m0, m1, g0, g1, slot_src, aux);
    }
    __syncthreads();
    // plane bases advance by one batch entry (C planes) when the row walk wraps: no 64-bit products in the loop
    const int64_t plane0 = static_cast<int64_t>(wi.n0) * p.C + wi.c;
    const S *xp = static_cast<const S *>(p.x) + plane0 * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + plane0 * p.o_plane;
    S *gxp = static_cast<S *>(p.out) + plane0 * p.x_plane;
    const int64_t xstep = static_cast<int64_t>(p.C) * p.x_plane, gstep = static_cast<int64_t>(p.C) * p.o_plane;
    if constexpr (kPrefetchPool) Stager::pool_load(NX, gp, slot_src, pk, aux, pregs);  // pooled values of the first step
    while (nl < wi.nn) {
        const int a = fdiv(r0, p.d_dim1);
        const int b0 = r0 - a * S1;
        const int Rn = step_len(r0);
        const int *ss = slot_src + tb * NS;
        char *tile = tile0 + (TILES == 2 ? buf * p.tile_bytes : 0);
        if constexpr (kPrefetchPool) {
            Stager::dma_x_decoded(NX, xp, ss, tile, pk);
            Stager::pool_expand(pregs, tile);
        } else if constexpr (DEC && POOL) Stager::issue_dma_decoded_pool(NX, xp, gp, ss, tile, pk, aux);
        else if constexpr (DEC) Stager::issue_dma_decoded(NX, xp, gp, ss, tile, pk);
        else Stager::issue_dma(p, R, xp, gp, ss, tile, aux);
        int nl2 = nl, r2 = r0 + Rn;
        if (r2 >= row_end) { r2 = wi.row0; ++nl2; }
        if (nl2 < wi.nn) {
            const int a2 = fdiv(r2, p.d_dim1);
            Stager::make_slots(p, R, a2, r2 - a2 * S1, step_len(r2), m0, m1, g0, g1, slot_src + ((tb + 1) % NT) * NS, aux);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (kPrefetchPool) {  // request the next step's pooled values; they arrive behind this step's compute
            if (nl2 < wi.nn) Stager::pool_load(NX, nl2 != nl ? gp + gstep : gp, slot_src + ((tb + 1) % NT) * NS, pk, aux, pregs);
        }
        if (worker && tr < Rn) {
            Chunk<S, E> res;
            // phase 1: grad_x (its corner values are dead before the x corners are read: fewer live registers)
            if constexpr (ACTIVE) {
                CT gv[NC][E + 1];
                lds_corners<T, ND>(tile, RB, R, NX + NG, ss, tr, gm, gv);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    CT v[1 << ND];
#pragma unroll
                    for (int q = 0; q < (1 << ND); ++q) v[q] = gv[q & (NC - 1)][e + (q >> (ND - 1))];
                    res.e[e] = narrow<T>(interp_t<T, ND>(v, dw));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (SCAT) {
                // the staged grad_out row b0 + tr, read through the column map, IS grad_x row scat_row (below)
                S graw[E + 1];
                lds_read_row<S, E>(tile + (NX + tr) * RB, true, gm, graw);
#pragma unroll
                for (int e = 0; e < E; ++e) res.e[e] = graw[e];
            } else {
                S graw[E + 1];
                lds_read_row<S, E>(tile + (NX + NG + tr) * RB, ss[NX + NG + tr] >= 0, gm, graw);
#pragma unroll
                for (int e = 0; e < E; ++e) res.e[e] = graw[e];
            }
            // phase 2: weight-gradient sums from the x corners and the incoming gradient
            CT xv[NC][E + 1];
            lds_corners<T, ND>(tile, RB, R, 0, ss, tr, xm, xv);
            const S *tg = reinterpret_cast<const S *>(tile + (NX + tr) * RB);
            Chunk<S, E> gch;
            __builtin_memcpy(gch.e, __builtin_assume_aligned(tg + ji, 16), 16);
            CT part[NDIFF];  // this chunk's sums, in the compute type (E terms each)
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) part[i] = CT(0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[1 << ND], df[NDIFF];
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) v[q] = xv[q & (NC - 1)][e + (q >> (ND - 1))];
                corner_diffs<ND, CT>(v, df);
                const CT gval = widen<T>(gch.e[e]);
#pragma unroll
                for (int i = 0; i < NDIFF; ++i) part[i] = fma_ct(gval, df[i], part[i]);  // one rounding per term: fewer instructions, not less accurate
            }
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) dsum[i] += static_cast<double>(part[i]);
            if constexpr (SCAT) {
                // grad_x row whose source is grad_out row b: periodic padding is a permutation (the x map is the inverse
                // of the grad map); otherwise b - shift when that is a row (rows [0, S1) of grad_out map to themselves),
                // and the rows left over at one end of the plane are written by the tail pass below
                const int b = b0 + tr;
                const int brow = p.pad == 2 ? m1[b] : b - scat_shift;
                if (brow >= 0 && brow < S1) store_chunk<S, E>(gxp + static_cast<int64_t>(brow) * S2 + ji, res);
            } else {
                store_chunk<S, E>(gxp + static_cast<int64_t>(a * S1 + b0 + tr) * S2 + ji, res);
            }
        }
        if (TILES == 1) __syncthreads();  // one tile: it is overwritten by the next step
        // (two tiles: the next step writes the other tile, last read before this step's barrier, and the slot table
        //  it uses was completed before that barrier too -- it is rewritten only after the next barrier)
        if (nl2 != nl) {
            xp += xstep;
            gp += gstep;
            gxp += xstep;
        }
        nl = nl2;
        r0 = r2;
        buf ^= 1;
        tb = (tb + 1) % NT;
    }
    if constexpr (SCAT) {
        // tail pass: the grad_x rows no grad_out row maps to by the plain shift -- |shift| rows at one end of every plane
        // (fill for zeros padding, clamped / reflected source rows otherwise); the workgroup whose band holds them
        // reads their sources straight from memory (a few rows per plane)
        if (p.pad != 2 && scat_shift != 0) {
            const int e0 = scat_shift > 0 ? max(S1 - scat_shift, 0) : 0, e1 = scat_shift > 0 ? S1 : min(-scat_shift, S1);
            const int lo = max(e0, wi.row0), hi = min(e1, row_end);
            bool gcontig = true;
#pragma unroll
            for (int e = 0; e < E; ++e) gcontig = gcontig && gm.cm[e] >= 0 && gm.cm[e] == gm.cm[0] + e;
            for (int pl = 0; pl < wi.nn; ++pl) {
                const S *gpl = static_cast<const S *>(p.go) + (plane0 + static_cast<int64_t>(pl) * p.C) * p.o_plane;
                S *gxl = static_cast<S *>(p.out) + (plane0 + static_cast<int64_t>(pl) * p.C) * p.x_plane;
                for (int bq = lo + tr; worker && bq < hi; bq += R) {
                    const int src = g1[bq];
                    Chunk<S, E> res;
                    S zero;
                    __builtin_memset(&zero, 0, sizeof(S));
                    if (src < 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) res.e[e] = zero;
                    } else if (gcontig) {
                        res = load_chunk<S, E>(gpl + static_cast<int64_t>(src) * S2 + gm.cm[0]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) res.e[e] = gm.cm[e] >= 0 ? gpl[static_cast<int64_t>(src) * S2 + gm.cm[e]] : zero;
                    }
                    store_chunk<S, E>(gxl + static_cast<int64_t>(bq) * S2 + ji, res);
                }
            }
        }
    }
    double acc[3] = {0.0, 0.0, 0.0};
    {
        const double dwd[3] = {static_cast<double>(dw[0]), static_cast<double>(dw[1]), static_cast<double>(dw[2])};
        blend_diffs<ND>(dsum, dwd, acc);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double t = block_sum(acc[s], scratch);
        if (threadIdx.x == 0) p.partials[(static_cast<size_t>(wi.pidx) * p.C + wi.c) * 3 + s] = t;
    }
}

// Gather forward through the same staging (SSL forward of 2/4/8-byte element types): R source rows per step,
// aligned LDS-DMA, shifted columns read from LDS.  Used for 16-bit rows, where 2-byte-aligned 16-byte global
// loads are slow (C5: 1.60 -> see DESIGN.md), and available for every eligible shape through tuning knob 2.
template <int ESIZE>
__global__ __launch_bounds__(kThreads) void plane_gather_forward_lds(const PlaneParams p) {
    using R_t = typename raw_t<ESIZE>::type;
    constexpr int E = 16 / ESIZE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int R = p.RPS;
    const int RB = p.S[2] * ESIZE;  // source row bytes, a multiple of 16
    char *tile = smem + 64;
    int *maps = reinterpret_cast<int *>(smem + 64 + p.tile_bytes);
    const int *m0 = maps, *m1 = maps + p.S[0] + 1, *m2 = m1 + p.S[1] + 1;
    int *slot_src = maps + p.S[0] + p.S[1] + p.S[2] + 3;  // two tables of R entries

    const WorkItem wi = decode_block(p);
    int64_t sh[3];
    gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(wi.c) * p.nd, p.wcol, sh);
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d] = p.wcol[d] >= 0 ? sh[d] : 0;
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    __syncthreads();

    const int O1 = p.O[1], O2 = p.O[2], S1 = p.S[1], S2 = p.S[2];
    const int tr = threadIdx.x / p.CW, tc = threadIdx.x - tr * p.CW;
    const bool worker = tr < R;
    const int jo = tc * E;
    const ColState<E> cs = make_colstate<E>(m2, jo + p.L[2], worker && (jo + E <= O2), p.lds_affine != 0);
    const R_t fill = static_cast<R_t>(p.fill);
    const int row_end = wi.row0 + wi.nrows;
    const int pieces = R * static_cast<int>(p.xppr);
    auto step_len = [&](int r0) {
        const int b0 = r0 - fdiv(r0, p.d_dim1) * O1;
        return min(R, min(O1 - b0, row_end - r0));
    };
    auto make_slots = [&](int r0, int Rn, int *tab) {
        const int k = threadIdx.x;
        if (k < R) {
            int src = -1;
            if (k < Rn) {
                const int a = fdiv(r0, p.d_dim1);
                const int b = r0 - a * O1 + k;
                const int ra = m0[a + p.L[0]], rb = m1[b + p.L[1]];
                if (ra >= 0 && rb >= 0) src = (ra * S1 + rb) * S2;
            }
            tab[k] = src;
        }
    };
    int nl = 0, r0 = wi.row0, buf = 0;
    make_slots(r0, step_len(r0), slot_src);
    // which pieces a thread moves is the same in every step: decode (slot, column piece) once (up to kPieces per thread)
    constexpr int kPieces = 2;
    const bool decoded = pieces <= kPieces * kThreads && p.xppr <= 256;
    int pk[kPieces];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        const int slot = fdiv(q, p.d_xppr);
        pk[k] = q < pieces ? slot * 256 + (q - slot * static_cast<int>(p.xppr)) : -1;
    }
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    __syncthreads();
    const int64_t plane0 = static_cast<int64_t>(wi.n0) * p.C + wi.c;
    const R_t *xp = static_cast<const R_t *>(p.x) + plane0 * p.x_plane;
    R_t *op = static_cast<R_t *>(p.out) + plane0 * p.o_plane;
    const int64_t xstep = static_cast<int64_t>(p.C) * p.x_plane, ostep = static_cast<int64_t>(p.C) * p.o_plane;
    while (nl < wi.nn) {
        const int Rn = step_len(r0);
        const int *ss = slot_src + buf * R;
        if (decoded) {
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                if (pk[k] >= 0) {
                    const int src = ss[pk[k] >> 8];
                    if (src >= 0) {
                        const R_t *g = xp + src + (pk[k] & 255) * E;
                        char *dst_wave = tile + (k * kThreads + wave * 64) * 16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
                    }
                }
            }
        } else {
            for (int q0 = 0; q0 < pieces; q0 += kThreads) {
                const int q = q0 + threadIdx.x;
                if (q < pieces) {
                    const int slot = fdiv(q, p.d_xppr);
                    const int j = q - slot * static_cast<int>(p.xppr);
                    const int src = ss[slot];
                    if (src >= 0) {
                        const R_t *g = xp + src + j * E;
                        char *dst_wave = tile + (q0 + (threadIdx.x & ~63)) * 16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
                    }
                }
            }
        }
        int nl2 = nl, r2 = r0 + Rn;
        if (r2 >= row_end) { r2 = wi.row0; ++nl2; }
        if (nl2 < wi.nn) make_slots(r2, step_len(r2), slot_src + (buf ^ 1) * R);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (worker && tr < Rn && jo + E <= O2) {
            R_t raw[E + 1];
            const bool valid = ss[tr] >= 0;
            lds_read_row<R_t, E>(tile + tr * RB, valid, cs, raw);
            Chunk<R_t, E> res;
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = (valid && cs.cm[e] >= 0) ? raw[e] : fill;
            store_chunk<R_t, E>(op + static_cast<int64_t>(r0 + tr) * O2 + jo, res);
        }
        __syncthreads();
        if (nl2 != nl) {
            xp += xstep;
            op += ostep;
        }
        nl = nl2;
        r0 = r2;
        buf ^= 1;
    }
}

// =====================================================================================================
// Host side: launch planning
// =====================================================================================================
bool contiguous(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    // strides in normalised order N, C, d0, d1, inner; size-1 dims may carry any stride
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

// diagnostics knobs (shiftnd_set_tuning): 0 = min workgroups wanted by the forward kernels (0 = automatic: 8192 for
// the gather forward, 2048 otherwise), 1 = target bytes per workgroup,
// 2 = gather-forward unroll (fp32/16-byte variant only), 6 = XCD-contiguous workgroup ids, 7 = workgroups
// wanted by the backward kernels (0 = automatic, see backward_min_wgs), 3 = backward / active-forward kernel: 2 LDS-staged where it applies (default), 1 direct global loads;
// 4 = LDS tiles of the backward kernel: 1 automatic, 2 two tiles + one barrier per step, 3 one tile; 5 = 1: no affine dword reads
thread_local int g_tune[8] = {0, 128 * 1024, 4, 2, 1, 0, 1, 0};

struct Plan {
    int V, cpr, CW, RPS, CP, ppw, groups, bands, rows_per_band, rows;
    size_t lds;
    unsigned grid;
};

// rows/inner: iteration space of one plane; esize: element bytes; V: chunk bytes
// row_unroll > 0 (gather forward): the workgroup walks ppw * rows super-rows in iterations of RPS * row_unroll; ppw is
// nudged (within -25 % .. +25 %) to the value whose last iteration wastes the fewest lanes
Plan make_plan(const Geometry &g, int64_t rows, int64_t inner, int esize, int V, int map_entries, int64_t min_wgs_override = 0,
               int row_unroll = 0) {
    Plan pl;
    pl.V = V;
    pl.rows = static_cast<int>(rows);
    pl.cpr = static_cast<int>(inner * esize / V);
    if (pl.cpr < 1) pl.cpr = 1;  // only reachable for ineligible geometries (workspace sizing)
    pl.CW = pl.cpr < kThreads ? pl.cpr : kThreads;
    pl.RPS = kThreads / pl.CW;
    pl.CP = (pl.cpr + pl.CW - 1) / pl.CW;
    const int64_t plane_bytes = rows * inner * esize;
    const int64_t min_wgs = min_wgs_override > 0 ? min_wgs_override : (g_tune[0] > 0 ? g_tune[0] : 2048);
    int64_t ppw = g_tune[1] / (plane_bytes > 0 ? plane_bytes : 1);
    if (ppw < 1) ppw = 1;
    if (ppw > g.N) ppw = g.N;
    auto ngroups = [&](int64_t q) { return (g.N + q - 1) / q; };
    while (ppw > 1 && g.C * ngroups(ppw) < min_wgs) ppw = (ppw + 1) / 2;
    if (row_unroll > 0 && ppw > 1) {
        const int64_t it = static_cast<int64_t>(pl.RPS) * row_unroll;
        auto waste = [&](int64_t q) { return static_cast<double>((q * rows + it - 1) / it * it) / static_cast<double>(q * rows); };
        int64_t best = ppw;
        for (int64_t q = ppw - ppw / 4; q <= ppw + ppw / 4 && q <= g.N; ++q)
            if (q >= 1 && waste(q) < waste(best) - 1e-9) best = q;
        ppw = best;
    }
    pl.ppw = static_cast<int>(ppw);
    pl.groups = static_cast<int>(ngroups(ppw));
    // few, large planes: cut each plane into row bands so that >= ~2048 workgroups exist
    int64_t bands = 1;
    const int64_t wgs = g.C * pl.groups;
    if (wgs < min_wgs && ppw == 1) {
        bands = (min_wgs + wgs - 1) / wgs;
        const int64_t min_rows = static_cast<int64_t>(pl.RPS) * 4;  // at least 4 row steps per band
        const int64_t max_bands = rows / (min_rows > 0 ? min_rows : 1);
        if (bands > max_bands) bands = max_bands;
        if (bands < 1) bands = 1;
    }
    pl.rows_per_band = static_cast<int>((rows + bands - 1) / bands);
    if (bands > 1)  // whole row steps per band (a ragged last step wastes lanes: 56 rows at 9 rows/step = 89 %)
        pl.rows_per_band = ((pl.rows_per_band + pl.RPS - 1) / pl.RPS) * pl.RPS;
    pl.bands = static_cast<int>((rows + pl.rows_per_band - 1) / pl.rows_per_band);
    pl.lds = static_cast<size_t>(map_entries) * sizeof(int);
    pl.grid = static_cast<unsigned>(g.C * pl.groups * pl.bands);
    return pl;
}

void fill_params(PlaneParams &p, const Geometry &g, const Plan &pl, int64_t dim1) {
    p.N = static_cast<int>(g.N);
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
    p.ppw = pl.ppw;
    p.groups = pl.groups;
    p.bands = pl.bands;
    p.rows_per_band = pl.rows_per_band;
    p.cpr = pl.cpr;
    p.CW = pl.CW;
    p.RPS = pl.RPS;
    p.CP = pl.CP;
    p.rows = pl.rows;
    p.d_rows = make_fastdiv(static_cast<uint32_t>(pl.rows_per_band));
    // With a wrapping / clamping padding every wave holds edge chunks whose maps are not affine, so it runs the affine
    // AND the per-element read path; for the 3-D kernels (8 + 8 staged rows per chunk) per-element reads alone are
    // cheaper (C3 with reflect padding: backward 0.82 -> 0.75 ms, forward 0.31 -> 0.30 ms; 2-D: no difference)
    p.lds_affine = g_tune[5] != 1 && !(g.pad != 0 && g.nd == 3 && g_tune[5] != 2);
    p.xcd_blocks = (g_tune[6] && pl.grid % 8 == 0) ? pl.grid / 8 : 0;
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(pl.cpr));
    p.d_dim1 = make_fastdiv(static_cast<uint32_t>(dim1));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_groups = make_fastdiv(static_cast<uint32_t>(pl.groups));
    for (int d = 0; d < 3; ++d) {
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
        p.d_perO[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.O[d], g.pad)));
    }
    for (int d = 0; d < 3; ++d) {
        p.K[d] = g.K[d] > 0 ? static_cast<int>(g.K[d]) : 1;
        p.P[d] = g.K[d] > 0 ? static_cast<int>(g.P[d]) : p.O[d];
        p.d_k[d] = make_fastdiv(static_cast<uint32_t>(p.K[d]));
    }
    p.d_p2 = make_fastdiv(static_cast<uint32_t>(p.P[2] > 0 ? p.P[2] : 1));
    if (g.K[0] > 0) p.o_plane = g.P[0] * g.P[1] * g.P[2];  // pooled calls: the output / incoming gradient is the pooled tensor
}

bool common_eligible(const Geometry &g) {
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe >= (1LL << 30) || oe >= (1LL << 30)) return false;           // 32-bit in-plane offsets
    if (g.N * g.C >= (1LL << 31) || g.N >= (1LL << 30) || g.C >= (1LL << 30)) return false;
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

template <int ESIZE, int V>
void launch_gather(const PlaneParams &p, const Plan &pl, hipStream_t st) {
    // narrow chunks = short ragged rows = few super-rows per workgroup: a shallow unroll wastes fewer lanes in the last
    // iteration (C4, 448 super-rows over 36 row lanes: U = 4 -> 0.136 ms, U = 2 -> 0.124 ms)
    if constexpr (V < 16) {
        hipLaunchKernelGGL((plane_gather_forward<ESIZE, V, 2>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
        return;
    }
    // (row steps per thread: 4; the other unrolls knob 2 once chose for 4-byte elements are no longer built)
    if constexpr (V >= 16) hipLaunchKernelGGL((plane_gather_forward<ESIZE, V, 4>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
}

// eligibility + LDS size of the LDS-staged kernels: 2-D / 3-D, no crop, one column pass, slot table <= threads
bool lds_staged_ok(const PlaneParams &p, const Plan &pl, int esize, int slots, size_t *lds_bytes, int *tile_bytes) {
    if ((p.nd != 2 && p.nd != 3) || pl.CP != 1) return false;
    // LDS-DMA moves aligned 16-byte pieces: row starts are multiples of 16 bytes from 16-byte aligned bases
    if (reinterpret_cast<uintptr_t>(p.x) % 16 != 0 || (p.go && reinterpret_cast<uintptr_t>(p.go) % 16 != 0)) return false;
    for (int d = 0; d < 3; ++d)
        if (p.L[d] != 0 || p.O[d] != p.S[d]) return false;  // no crop: output rows == input rows
    const size_t tile = static_cast<size_t>(slots) * p.S[2] * esize;
    const size_t total = 64 + tile + pl.lds + 2 * static_cast<size_t>(slots) * sizeof(int);
    if (total > 64 * 1024 || slots > kThreads) return false;
    *lds_bytes = total;
    *tile_bytes = static_cast<int>(tile);
    return true;
}

template <typename T>
int launch_active_forward(const PlaneParams &p_in, const Plan &pl, hipStream_t st) {
    const PlaneParams &p = p_in;
    // (the LDS-staged form, plane_active_forward_lds, is no longer built: every un-cropped problem of whole 16-byte rows it took is
    //  step_forward_lds's, walk_forward's or slide_forward's -- no default-routed shape in tools/route_census.py's 6 000 problems)
    note_kernel("plane_active_forward");
    switch (p.nd) {
    case 1: hipLaunchKernelGGL((plane_active_forward<T, 1>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    case 2: hipLaunchKernelGGL((plane_active_forward<T, 2>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    default: hipLaunchKernelGGL((plane_active_forward<T, 3>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    }
    return SHIFTND_OK;
}

template <typename T, bool ACTIVE>
void launch_backward_a(const PlaneParams &p_in, const Plan &pl, hipStream_t st) {
    PlaneParams p = p_in;
    // LDS-staged form where it applies: 3-D volumes (what the walk through the planes does not take -- fp64, rows wider than
    // half a workgroup pass, one-row planes ...).  The 2-D forms are no longer built: every dense un-cropped 2-D problem with
    // rows of at most 256 pieces is step_backward's (tools/route_census.py found no default-routed shape for them).
    if ((g_tune[3] == 2 || g_tune[3] == 3) && p.nd == 3) {
        size_t lds_bytes = 0;
        int tile_bytes = 0;
        const int slots = LdsTileShape<3, ACTIVE, true>::slots(pl.RPS);
        if (lds_staged_ok(p, pl, static_cast<int>(sizeof(typename T::S)), slots, &lds_bytes, &tile_bytes)) {
            p.tile_bytes = tile_bytes;
            note_kernel("plane_backward_lds");
            // two LDS tiles / one barrier per step pay when registers, not LDS, bound the resident workgroups: 16-bit dtypes
            // (C5 backward 2.46 -> 2.21 ms; fp32 C2 1.84 -> 1.96 ms: 4- / 8-byte elements keep one tile, and only that form
            // is built for them); knob 4 = 3 forces one tile for 16-bit data too
            constexpr bool kTwoTiles = sizeof(typename T::S) == 2;
            const bool two = kTwoTiles && g_tune[4] != 3 && lds_bytes + tile_bytes + slots * sizeof(int) <= 64 * 1024;
            const size_t lds2 = lds_bytes + tile_bytes + slots * sizeof(int);
            // few pieces per thread and step: the pre-decoded DMA form (knob 5 = 2 keeps the generic loop)
            const bool dec = g_tune[5] != 2 && LdsStager<T, 3, ACTIVE, true>::pieces_fit(pl.cpr, pl.RPS);
#define SHIFTND_BWD_LDS(TL, DECV, BYTES) \
    hipLaunchKernelGGL((plane_backward_lds<T, 3, ACTIVE, TL, false, DECV>), dim3(pl.grid), dim3(kThreads), BYTES, st, p)
            if constexpr (kTwoTiles) {
                if (two) {
                    if (dec) SHIFTND_BWD_LDS(2, true, lds2); else SHIFTND_BWD_LDS(2, false, lds2);
                    return;
                }
            }
            if (dec) SHIFTND_BWD_LDS(1, true, lds_bytes); else SHIFTND_BWD_LDS(1, false, lds_bytes);
#undef SHIFTND_BWD_LDS
            return;
        }
    }
    note_kernel("plane_backward");
    switch (p.nd) {
    case 1: hipLaunchKernelGGL((plane_backward<T, 1, ACTIVE>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    case 2: hipLaunchKernelGGL((plane_backward<T, 2, ACTIVE>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    default: hipLaunchKernelGGL((plane_backward<T, 3, ACTIVE>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    }
}

template <typename T, bool ACTIVE>
void launch_backward_pool(const PlaneParams &p_in, const Plan &pl, hipStream_t st) {
    PlaneParams p = p_in;
    // LDS-staged form where it applies (pooled rows are expanded into the tile).  The 3-D interpolating form is not built:
    // shiftnd_backward_pooled hands those volumes to the walk or reports SHIFTND_ERR_NOT_FUSED (measured slower than
    // avg_pool backward + shiftnd_backward)
    constexpr bool k3d = !ACTIVE;
    if (g_tune[3] == 2 && (p.nd == 2 || (p.nd == 3 && k3d))) {
        size_t lds_bytes = 0;
        int tile_bytes = 0;
        const int slots = p.nd == 3 ? LdsTileShape<3, ACTIVE, true>::slots(pl.RPS) : LdsTileShape<2, ACTIVE, true>::slots(pl.RPS);
        if (lds_staged_ok(p, pl, static_cast<int>(sizeof(typename T::S)), slots, &lds_bytes, &tile_bytes)) {
            p.tile_bytes = tile_bytes;
            note_kernel("plane_backward_lds_pool");
            lds_bytes += 3 * slots * sizeof(int);  // the second set of slot tables (window counts)
            // tiles as in launch_backward_a: two for 16-bit data when they fit, one for 4- / 8-byte elements
            constexpr bool kTwoTiles = sizeof(typename T::S) == 2;
            const bool two = kTwoTiles && g_tune[4] != 3 && lds_bytes + tile_bytes + slots * sizeof(int) <= 64 * 1024;
            const size_t bytes = two ? lds_bytes + tile_bytes + slots * sizeof(int) : lds_bytes;
            // windows of 2 along the row and few pieces per thread: the pre-decoded form (default tile count only)
            constexpr int kDefTiles = kTwoTiles ? 2 : 1;
            const bool dec = g_tune[5] != 2 && p.K[2] == 2 && two == kTwoTiles &&
                             (p.nd == 3 ? LdsStager<T, 3, ACTIVE, true, true>::pieces_fit(pl.cpr, pl.RPS)
                                        : LdsStager<T, 2, ACTIVE, true, true>::pieces_fit(pl.cpr, pl.RPS));
#define SHIFTND_BWD_POOL(NDV, TL, DECV) \
    hipLaunchKernelGGL((plane_backward_lds<T, NDV, ACTIVE, TL, true, DECV>), dim3(pl.grid), dim3(kThreads), bytes, st, p)
            if (p.nd == 3) {
                if constexpr (k3d) {
                    if (dec) SHIFTND_BWD_POOL(3, kDefTiles, true);
                    else if (two) SHIFTND_BWD_POOL(3, kDefTiles, false);
                    else SHIFTND_BWD_POOL(3, 1, false);
                }
            } else {
                if (dec) SHIFTND_BWD_POOL(2, kDefTiles, true);
                else if (two) SHIFTND_BWD_POOL(2, kDefTiles, false);
                else SHIFTND_BWD_POOL(2, 1, false);
            }
#undef SHIFTND_BWD_POOL
            return;
        }
    }
    note_kernel("plane_backward_pool");
    switch (p.nd) {
    case 1: hipLaunchKernelGGL((plane_backward<T, 1, ACTIVE, true>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    case 2: hipLaunchKernelGGL((plane_backward<T, 2, ACTIVE, true>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    default: hipLaunchKernelGGL((plane_backward<T, 3, ACTIVE, true>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    }
}

template <typename T>
int launch_pool_forward(const PlaneParams &p, const Plan &pl, bool active, hipStream_t st) {
    note_kernel("plane_pool_forward");
    const bool smallk = p.K[0] <= 2 && p.K[1] <= 2 && p.K[2] <= 2;
#define SHIFTND_POOL_FWD(NDV) \
    if (active && smallk) hipLaunchKernelGGL((plane_pool_forward<T, NDV, true, true>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); \
    else if (active) hipLaunchKernelGGL((plane_pool_forward<T, NDV, true, false>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); \
    else if (smallk) hipLaunchKernelGGL((plane_pool_forward<T, NDV, false, true>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p); \
    else hipLaunchKernelGGL((plane_pool_forward<T, NDV, false, false>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    switch (p.nd) {
    case 1: SHIFTND_POOL_FWD(1) break;
    case 2: SHIFTND_POOL_FWD(2) break;
    default: SHIFTND_POOL_FWD(3) break;
    }
#undef SHIFTND_POOL_FWD
    return SHIFTND_OK;
}

template <typename T>
int launch_backward(const PlaneParams &p, const Plan &pl, bool active, void *gw, hipStream_t st, bool pool = false) {
    if (pool) {
        if (active) launch_backward_pool<T, true>(p, pl, st);
        else launch_backward_pool<T, false>(p, pl, st);
    } else if (active) launch_backward_a<T, true>(p, pl, st);
    else launch_backward_a<T, false>(p, pl, st);
    reduce_weight_grads_of<T>(p.partials, pl.groups * pl.bands, p.C, p.nd, gw, st);
    return SHIFTND_OK;
}

// Workgroups wanted by the backward kernels (knob 7 overrides).  Measured on MI355X (tools/kbench.py --knobs 7=...):
// the 2-D SSL kernel on 4-byte data (62 VGPRs, 8 workgroups per CU) is fastest with large planes cut into ~14-step
// row bands (C2: 65536 -> 1.71 ms, 16384 -> 1.83 ms); every other variant holds more registers, keeps fewer
// workgroups resident to hide its prologue behind, and wants few, long workgroups (8192: C5 fp16 2.04 -> 1.86 ms,
// C3 bf16 3-D active 0.79 -> 0.65 ms, fp32 3-D SSL 0.82 -> 0.66 ms, 2-D active fp32 2.06 -> 1.98 ms,
// N128 C512 56x56 fp32 0.59 -> 0.49 ms).
int64_t backward_min_wgs(const Geometry &g, int esize) {
    if (g_tune[7] > 0) return g_tune[7];
    const int64_t plane_bytes = g.S[0] * g.S[1] * g.S[2] * esize;
    if (g.K[0] > 0) return 8192;  // fused-pool backward (more registers): N64 C256 224x224 fp32 2.35 -> 2.13 ms
    if (!(g.nd <= 2 && !g.active && esize >= 4 && plane_bytes >= g_tune[1])) return 8192;
    // large planes, light kernel: row bands of >= 14 steps each, up to 65536 workgroups in all (C2 N64 C256: 4 bands;
    // N64 C64 224x224: 4 bands = 16384 workgroups, 0.43 ms where 65536 workgroups of 3.5 steps took 0.52 ms)
    const int64_t cpr = g.S[2] * esize / 16 > 0 ? g.S[2] * esize / 16 : 1;
    const int64_t rps = cpr < kThreads ? kThreads / cpr : 1;
    const int64_t steps = (g.S[0] * g.S[1] + rps - 1) / rps;
    const int64_t planes = g.N * g.C;
    int64_t bands = planes < 65536 ? 65536 / planes : 1;
    if (bands > steps / 14) bands = steps / 14;
    if (bands < 1) bands = 1;
    return planes * bands;
}

Plan backward_plan(const Geometry &g, int esize) {
    const int entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + g.O[0] + g.O[1] + g.O[2] + 6);
    // (a batch walk -- a workgroup keeps one step of rows and walks n, every step 51 MB ahead -- was measured 7 - 16 % slower
    //  than the band walk: DESIGN section 9)
    return make_plan(g, g.S[0] * g.S[1], g.S[2], esize, 16, entries, backward_min_wgs(g, esize));
}

}  // namespace

void plane_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 8) g_tune[knob] = value;
}

static bool lds_gather_wanted(const Geometry &g, int es, int V, const Plan &pl, const void *x) {
    if (!(V == 16 && es >= 2 && (g.S[2] * es) % 16 == 0 && pl.CP == 1 && reinterpret_cast<uintptr_t>(x) % 16 == 0)) return false;
    if (!((es == 2 && g_tune[2] == 4) || g_tune[2] == 16)) return false;
    const size_t total = 64 + static_cast<size_t>(pl.RPS) * g.S[2] * es + pl.lds + 2 * static_cast<size_t>(pl.RPS) * sizeof(int);
    return total <= 64 * 1024;
}

// workgroups wanted by the gather forward (knob 0 overrides): a light kernel, 8 workgroups per CU resident -- 2048
// workgroups are one round with no slack for uneven finish times (C4 quint8 N128 C512 56x56: 0.187 -> 0.134 ms)
static int64_t gather_min_wgs() { return g_tune[0] > 0 ? g_tune[0] : 8192; }

// true when plane_forward would take the LDS-staged gather kernel (the API then prefers it over the sweep kernel)
bool plane_forward_lds_gather(const Geometry &g, int dtype, const void *x, const void *out) {
    if ((g.active && dtype <= SHIFTND_BF16) || !plane_forward_eligible(g, dtype, x, out)) return false;
    const int es = dtype_size(dtype);
    const int V = gather_vector_bytes(g, es, out);
    const Plan pl = make_plan(g, g.O[0] * g.O[1], g.O[2], es, V, static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3), gather_min_wgs(),
                              V < 16 ? 2 : 4);
    return lds_gather_wanted(g, es, V, pl, x);
}

bool plane_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    (void)x;
    if (!common_eligible(g)) return false;
    if (g.S[0] + g.S[1] + g.S[2] + 3 > kMaxMapEntries) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O)) return false;
    const bool interpolating = g.active && dtype <= SHIFTND_BF16;
    if (interpolating) {
        const int es = dtype_size(dtype);
        if ((g.O[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return false;
    }
    return true;
}

int plane_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp,
                  uint64_t fill_bits, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const bool interpolating = g.active && dtype <= SHIFTND_BF16;
    const int entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    PlaneParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = fill_bits;
    if (interpolating) {
        if (slide_forward_eligible(g, dtype, x, out)) return slide_forward(g, dtype, x, w, out, st);
        const Plan pl = make_plan(g, g.O[0] * g.O[1], g.O[2], es, 16, entries);
        fill_params(p, g, pl, g.O[1]);
        switch (dtype) {
        case SHIFTND_F32: return launch_active_forward<f32_t>(p, pl, st);
        case SHIFTND_F64: return launch_active_forward<f64_t>(p, pl, st);
        case SHIFTND_F16: return launch_active_forward<f16_t>(p, pl, st);
        default: return launch_active_forward<bf16_t>(p, pl, st);
        }
    }
    if (bytes_forward_eligible(g, dtype, x, out)) return bytes_forward(g, x, w, wkind, wzp, fill_bits, out, st);
    if (bytes_block_forward_eligible(g, dtype, x, out)) return bytes_block_forward(g, x, w, wkind, wzp, fill_bits, out, st);
    const int V = gather_vector_bytes(g, es, out);
    note_kernel("plane_gather_forward");
    const Plan pl = make_plan(g, g.O[0] * g.O[1], g.O[2], es, V, entries, gather_min_wgs(), V < 16 ? 2 : 4);
    fill_params(p, g, pl, g.O[1]);
    // LDS-staged gather: 16-bit rows by default (knob 2 == 4), every eligible element size with knob 2 == 16
    if (lds_gather_wanted(g, es, V, pl, x)) {
        const size_t tile = static_cast<size_t>(pl.RPS) * g.S[2] * es;
        const size_t total = 64 + tile + pl.lds + 2 * static_cast<size_t>(pl.RPS) * sizeof(int);
        {
            p.tile_bytes = static_cast<int>(tile);
            p.xppr = static_cast<unsigned>(g.S[2] * es / 16);
            p.d_xppr = make_fastdiv(p.xppr);
            note_kernel("plane_gather_forward_lds");
            if (es == 2) hipLaunchKernelGGL((plane_gather_forward_lds<2>), dim3(pl.grid), dim3(kThreads), total, st, p);
            else if (es == 4) hipLaunchKernelGGL((plane_gather_forward_lds<4>), dim3(pl.grid), dim3(kThreads), total, st, p);
            else hipLaunchKernelGGL((plane_gather_forward_lds<8>), dim3(pl.grid), dim3(kThreads), total, st, p);
            return SHIFTND_OK;
        }
    }
#define SHIFTND_GATHER_CASE(ES, VV) \
    if (es == ES && V == VV) { launch_gather<ES, VV>(p, pl, st); return SHIFTND_OK; }
    SHIFTND_GATHER_CASE(1, 16) SHIFTND_GATHER_CASE(1, 8) SHIFTND_GATHER_CASE(1, 4) SHIFTND_GATHER_CASE(1, 1)
    SHIFTND_GATHER_CASE(2, 16) SHIFTND_GATHER_CASE(2, 8) SHIFTND_GATHER_CASE(2, 4) SHIFTND_GATHER_CASE(2, 2)
    SHIFTND_GATHER_CASE(4, 16) SHIFTND_GATHER_CASE(4, 8) SHIFTND_GATHER_CASE(4, 4)
    SHIFTND_GATHER_CASE(8, 16) SHIFTND_GATHER_CASE(8, 8)
#undef SHIFTND_GATHER_CASE
    return SHIFTND_ERR_UNSUPPORTED_DTYPE;
}

bool plane_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    (void)go;
    (void)x;
    if (dtype > SHIFTND_BF16 || !common_eligible(g)) return false;
    if (g.S[0] + g.S[1] + g.S[2] + g.O[0] + g.O[1] + g.O[2] + 6 > kMaxMapEntries) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O) || !contiguous(g.gs, g.N, g.C, g.S))
        return false;
    const int es = dtype_size(dtype);
    if ((g.S[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(gx) % 16 != 0) return false;
    return true;
}

size_t plane_backward_workspace(const Geometry &g, int dtype) {
    const Plan pl = backward_plan(g, dtype_size(dtype));
    const size_t own = static_cast<size_t>(pl.groups) * pl.bands * static_cast<size_t>(g.C) * 3 * sizeof(double);
    const size_t slide = g.K[0] > 0 ? 0 : slide_backward_workspace(g, dtype);  // (the fused-pool calls never slide)
    const size_t step = std::max(step_backward_workspace(g, dtype), g.K[0] > 0 ? span_backward_pooled_workspace(g, dtype)
                                                                                : std::max(walk16_backward_workspace(g, dtype), span_backward_workspace(g, dtype)));
    const size_t m = own > slide ? own : slide;
    return m > step ? m : step;
}

// ---- fused shift + average pool (contiguous tensors; g.K / g.P set) -----------------------------------------
bool plane_pool_forward_eligible(const Geometry &g, int dtype) {
    if (dtype > SHIFTND_BF16 || !common_eligible(g)) return false;
    return g.S[0] + g.S[1] + g.S[2] + 3 <= kMaxMapEntries;
}

int plane_pool_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const int entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    PlaneParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = dtype;
    const Plan pl = make_plan(g, g.P[0] * g.P[1], g.P[2], es, es, entries);
    fill_params(p, g, pl, g.P[1]);
    switch (dtype) {
    case SHIFTND_F32: return launch_pool_forward<f32_t>(p, pl, g.active != 0, st);
    case SHIFTND_F64: return launch_pool_forward<f64_t>(p, pl, g.active != 0, st);
    case SHIFTND_F16: return launch_pool_forward<f16_t>(p, pl, g.active != 0, st);
    default: return launch_pool_forward<bf16_t>(p, pl, g.active != 0, st);
    }
}

bool plane_pool_backward_eligible(const Geometry &g, int dtype, const void *gx) {
    if (dtype > SHIFTND_BF16 || !common_eligible(g)) return false;
    if (g.S[0] + g.S[1] + g.S[2] + g.O[0] + g.O[1] + g.O[2] + 6 > kMaxMapEntries) return false;
    const int es = dtype_size(dtype);
    return (g.S[2] * es) % 16 == 0 && reinterpret_cast<uintptr_t>(gx) % 16 == 0;
}

extern thread_local int g_step_tune[5];   // knobs 32..35 / 38 (shiftnd_step.hip)

int plane_pool_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                        void *workspace, hipStream_t st) {
    if (step_backward_pooled_eligible(g, dtype, go, x, gx)) return step_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    // cropped windows and the interpolating shift: crop_backward<.., POOL> (round 6; knob 35 bit 6 keeps the band walk)
    if (!(g_step_tune[3] & 64) && span_backward_pooled_eligible(g, dtype, go, x, gx)) return span_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    const Plan pl = backward_plan(g, dtype_size(dtype));
    PlaneParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_params(p, g, pl, g.S[1]);
    switch (dtype) {
    case SHIFTND_F32: return launch_backward<f32_t>(p, pl, g.active != 0, gw, st, true);
    case SHIFTND_F64: return launch_backward<f64_t>(p, pl, g.active != 0, gw, st, true);
    case SHIFTND_F16: return launch_backward<f16_t>(p, pl, g.active != 0, gw, st, true);
    default: return launch_backward<bf16_t>(p, pl, g.active != 0, gw, st, true);
    }
}

int plane_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st) {
    if (walk16_backward_eligible(g, dtype, go, x, gx)) return walk16_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    if (step_backward_eligible(g, dtype, go, x, gx)) return step_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    if (span_backward_eligible(g, dtype, go, x, gx)) return span_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    if (slide_backward_eligible(g, dtype, go, x, gx)) return slide_backward(g, dtype, go, x, w, gx, gw, workspace, st);
    const Plan pl = backward_plan(g, dtype_size(dtype));
    PlaneParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_params(p, g, pl, g.S[1]);
    switch (dtype) {
    case SHIFTND_F32: return launch_backward<f32_t>(p, pl, g.active != 0, gw, st);
    case SHIFTND_F64: return launch_backward<f64_t>(p, pl, g.active != 0, gw, st);
    case SHIFTND_F16: return launch_backward<f16_t>(p, pl, g.active != 0, gw, st);
    default: return launch_backward<bf16_t>(p, pl, g.active != 0, gw, st);
    }
}

// ---- rows that are not whole 16-byte pieces, behind every chunk kernel (round 5: 3-D volumes; round 6: the rest of the tail) -----
// 16 x 28 x 28 bf16 (56-byte rows), 8 x 56 x 62 fp32 ...: volumes beyond the small-plane kernels' 16 KiB.  The route census found
// them on the strided fallback (one thread per element, 64-bit index arithmetic, one workgroup per (n, c)): 0.02 - 0.5 TB/s.  The
// direct-load plane kernels serve them with chunks of 4 bytes (two 16-bit elements / one fp32) or 8 (one fp64): padding maps in LDS,
// row bands across workgroups, coalesced element-aligned loads and stores.
// Round 6 (route census of 6 000 problems: 164 backward / 136 interpolating-forward calls still on the fallback): rows of an ODD
// number of 16-bit elements move one element per thread (VB = 2); 1-D rows cut by a window that the flat stream declines (rows of
// 5 - 28 elements: more than 72 planes per 4 KiB step) and -- forward -- 2-D windows on rows too long for the row-span kernels run
// the same kernels with one element per thread.
// (tensors at an element-aligned storage offset -- a view into a larger buffer -- are declined by every 16-byte-piece family;
//  they end here too, whatever their rows: element-wide chunks need the element's own alignment only)
static int ragged_vector_bytes(int es, int64_t row_elems, int nd, const void *a, const void *b, const void *c) {
    const bool dwords = (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) % 4 == 0;
    if (es == 2) return (row_elems % 2 == 0 && nd == 3 && dwords) ? 4 : 2;
    return es;
}

bool plane_ragged_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (dtype > SHIFTND_BF16 || !g.active || g.nd < 1 || g.nd > 3 || g.K[0] > 0 || !common_eligible(g)) return false;
    if (g.S[0] + g.S[1] + g.S[2] + 3 > kMaxMapEntries) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O)) return false;
    // (rows of whole pieces are plane_forward's, earlier in the route -- unless a pointer is not 16-byte aligned)
    return reinterpret_cast<uintptr_t>(out) % dtype_size(dtype) == 0 && reinterpret_cast<uintptr_t>(x) % dtype_size(dtype) == 0;
}

int plane_ragged_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const int entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    PlaneParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = dtype;
    const int vb = ragged_vector_bytes(es, g.O[2], g.nd, x, out, out);
    const Plan pl = make_plan(g, g.O[0] * g.O[1], g.O[2], es, vb, entries);
    fill_params(p, g, pl, g.O[1]);
    note_kernel("plane_active_forward_ragged");
    const dim3 grid(pl.grid), block(kThreads);
#define SHIFTND_RAGGED_FWD(TT, VB1, VB3) \
    if (g.nd == 1) hipLaunchKernelGGL((plane_active_forward<TT, 1, VB1>), grid, block, pl.lds, st, p); \
    else if (g.nd == 2) hipLaunchKernelGGL((plane_active_forward<TT, 2, VB1>), grid, block, pl.lds, st, p); \
    else if (vb == VB1) hipLaunchKernelGGL((plane_active_forward<TT, 3, VB1>), grid, block, pl.lds, st, p); \
    else hipLaunchKernelGGL((plane_active_forward<TT, 3, VB3>), grid, block, pl.lds, st, p);
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_RAGGED_FWD(f32_t, 4, 4) break;
    case SHIFTND_F64: SHIFTND_RAGGED_FWD(f64_t, 8, 8) break;
    case SHIFTND_F16: SHIFTND_RAGGED_FWD(f16_t, 2, 4) break;
    default: SHIFTND_RAGGED_FWD(bf16_t, 2, 4) break;
    }
#undef SHIFTND_RAGGED_FWD
    return SHIFTND_OK;
}

// (vb: the chunk width; the plan's record count is largest for the narrowest chunk, so the workspace is sized for vb = es)
static Plan ragged_backward_plan(const Geometry &g, int es, int vb) {
    const int entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + g.O[0] + g.O[1] + g.O[2] + 6);
    return make_plan(g, g.S[0] * g.S[1], g.S[2], es, vb, entries, backward_min_wgs(g, es));
}

bool plane_ragged_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (dtype > SHIFTND_BF16 || g.nd < 1 || g.nd > 3 || g.K[0] > 0 || !common_eligible(g)) return false;
    if (g.S[0] + g.S[1] + g.S[2] + g.O[0] + g.O[1] + g.O[2] + 6 > kMaxMapEntries) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O) || !contiguous(g.gs, g.N, g.C, g.S)) return false;
    // (rows of whole pieces are plane_backward's, earlier in the route -- unless a pointer is not 16-byte aligned)
    const uintptr_t es = static_cast<uintptr_t>(dtype_size(dtype));
    return reinterpret_cast<uintptr_t>(gx) % es == 0 && reinterpret_cast<uintptr_t>(x) % es == 0 && reinterpret_cast<uintptr_t>(go) % es == 0;
}

size_t plane_ragged_backward_workspace(const Geometry &g, int dtype) {
    if (dtype > SHIFTND_BF16 || g.nd < 1 || g.nd > 3 || g.C < 1 || g.N < 1 || g.S[0] * g.S[1] * g.S[2] < 1) return 0;
    const int es = dtype_size(dtype);
    const Plan a = ragged_backward_plan(g, es, es), b = ragged_backward_plan(g, es, es == 2 ? 4 : es);
    const size_t recs = std::max(static_cast<size_t>(a.groups) * a.bands, static_cast<size_t>(b.groups) * b.bands);
    return recs * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

int plane_ragged_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                          void *workspace, hipStream_t st) {
    const int es = dtype_size(dtype);
    const int vb = ragged_vector_bytes(es, g.S[2], g.nd, go, x, gx);
    const Plan pl = ragged_backward_plan(g, es, vb);
    PlaneParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_params(p, g, pl, g.S[1]);
    note_kernel("plane_backward_ragged");
    const dim3 grid(pl.grid), block(kThreads);
    const bool active = g.active != 0;
#define SHIFTND_RAGGED_BWD_K(TT, NDV, VBV) \
    if (active) hipLaunchKernelGGL((plane_backward<TT, NDV, true, false, VBV>), grid, block, pl.lds, st, p); \
    else hipLaunchKernelGGL((plane_backward<TT, NDV, false, false, VBV>), grid, block, pl.lds, st, p);
#define SHIFTND_RAGGED_BWD(TT, VB1, VB3) \
    if (g.nd == 1) { SHIFTND_RAGGED_BWD_K(TT, 1, VB1) } \
    else if (g.nd == 2) { SHIFTND_RAGGED_BWD_K(TT, 2, VB1) } \
    else if (vb == VB1) { SHIFTND_RAGGED_BWD_K(TT, 3, VB1) } \
    else { SHIFTND_RAGGED_BWD_K(TT, 3, VB3) } \
    reduce_weight_grads_of<TT>(p.partials, pl.groups * pl.bands, p.C, p.nd, gw, st);
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_RAGGED_BWD(f32_t, 4, 4) break;
    case SHIFTND_F64: SHIFTND_RAGGED_BWD(f64_t, 8, 8) break;
    case SHIFTND_F16: SHIFTND_RAGGED_BWD(f16_t, 2, 4) break;
    default: SHIFTND_RAGGED_BWD(bf16_t, 2, 4) break;
    }
#undef SHIFTND_RAGGED_BWD
#undef SHIFTND_RAGGED_BWD_K
    return SHIFTND_OK;
}

}  // namespace shiftnd
