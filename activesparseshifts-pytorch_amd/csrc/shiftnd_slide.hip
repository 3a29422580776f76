// shiftnd_slide.hip -- sliding-window kernels for gfx950 (MI355X): the backward pass and the interpolating forward
// of 2-D / 3-D problems (contiguous tensors, no crop, rows made of whole 16-byte pieces).
//
// Why another family (measured on C3, bf16 3-D active backward, round 2 baseline: profiles/r02_c3_before_*): the
// step-tiled kernel of shiftnd_plane.hip re-reads every corner row for every output row (8 staged-row reads, 72
// widenings and 72 masks per 8-element chunk), stages 5.2 rows per output row, and waits for its own LDS-DMA before
// it computes.  Here
//   * a thread keeps ONE chunk column of ONE segment (a run of consecutive rows b of one plane a) and walks down the
//     rows: the rows it read for output row b (hb = 1 corners) are the hb = 0 corners of row b + 1 and stay in
//     registers -- already widened, and (3-D) already blended over the two a-planes.  Per output row a thread reads
//     NA new rows per tensor instead of 2 NA, and the interpolation is evaluated as the reference nests it
//     (interpolation.h:34-40: blend over dim0, then dim1, then the inner dim), so results are bit-identical to
//     interp_nd while every partial blend is computed once.
//   * 3-D: the segments of a workgroup are consecutive a-planes at the same rows b, so the "+1 along dim0" corner row
//     of segment s IS the row of segment s + 1: a step stages nseg + 1 rows per tensor for nseg output rows
//     (C3: 50 staged rows per 16 output rows; the step-tiled kernel staged 94 per 18).
//   * staging goes global -> registers -> LDS, one step ahead: the loads of step t + 1 are in flight while step t is
//     computed (LDS-DMA would make hipcc wait vmcnt(0) before the first LDS read of the compute phase).  Two LDS
//     tiles alternate, one barrier per step.
//   * rows are written to LDS displaced by 0..3 dwords (the channel's inner shift is the same for every thread of the
//     workgroup), so every thread's window of E + 1 shifted columns starts at a 16-byte boundary: one ds_read_b128 +
//     one ds_read_b32 per row, no bank conflicts between the lanes of a row; 16-bit types take one uniform funnel
//     shift (v_alignbit) for the odd half.  Every row has zeroed guard bytes on both sides, so with zeros padding the
//     columns outside the row read as 0 without a mask; chunks whose column map is not affine (edges of the wrapping
//     / clamping paddings) read element by element through the LDS column map.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/): backward kernels/shifts_kernels.h:222-327 with
// kernels/interpolation.h:3-61, forward kernels/shifts_kernels.h:156-220; weight preparation cpu/shifts_cpu.cpp:223-224,
// :242-244.  Roofline: HBM; backward 3*s bytes per element, forward 2*s.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kGuard = 32;          // zeroed bytes on each side of a staged row
constexpr int kFlushSteps = 4;      // fp32 partial sums go to the fp64 accumulators every kFlushSteps rows

struct SlideParams {
    const void *x;       // forward: input; backward: saved input
    const void *go;      // backward: incoming gradient
    void *out;           // forward: output; backward: grad_x
    const void *w;
    double *partials;    // backward: [workgroups per channel][C][3]
    int64_t plane;       // elements per (n, c) volume
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int cpr;             // 16-byte pieces per row
    int nseg;            // segments per workgroup
    int seg;             // rows per segment (band length); a workgroup runs seg + 1 steps
    int bands;           // bands per plane along dim1
    int agroups;         // 3-D: groups of nseg a-planes
    int inner;           // workgroups that share one (n, c): 3-D agroups * bands, 2-D 1
    int units;           // 2-D: N * bands (n, band) units per channel
    int pitch;           // LDS bytes per staged row (row bytes + 2 guards)
    int tile_bytes;
    int npieces;         // 16-byte pieces staged per step
    unsigned xcd_blocks;
    FastDiv d_cpr, d_bands, d_inner, d_C, d_per;
};

template <typename T> struct ElemTraits {
    using S = typename T::S;
    static constexpr int ES = sizeof(S);
    static constexpr int E = 16 / ES;
};

template <typename T> __device__ __forceinline__ typename T::C lerp_t(typename T::C a, typename T::C b, typename T::C x) {
    // one lerp of interp_t (shiftnd_common.hpp): the reference's mul + mul + add for fp32, mul + fma for 16-bit data
    if constexpr (sizeof(typename T::S) == 2) return lerp1_fused(a, b, x);
    else return lerp1(a, b, x);
}

// How a thread reads its E + 1 shifted columns of a staged row (one per tensor kind: x rows, grad_out rows).
struct RowRead {
    int woff;     // affine lanes: byte offset, from the row's slot base, of the 16-byte aligned window start
    int fboff;    // element-wise lanes: byte offset of source column 0 from the slot base (kGuard - displacement)
    int half;     // 16-bit types: 16 when the window starts at the odd half of its first dword (uniform), else 0
    bool affine;
};

// E + 1 raw elements as 5 dwords, first element in the low bits of t[0]
template <int ES> __device__ __forceinline__ void read_window(const char *p, int half, uint32_t (&t)[5]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(p, 16));
    const uint32_t d4 = *reinterpret_cast<const uint32_t *>(p + 16);
    if constexpr (ES == 2) {
        const uint32_t sh = static_cast<uint32_t>(half);
        t[0] = __builtin_amdgcn_alignbit(q.y, q.x, sh);
        t[1] = __builtin_amdgcn_alignbit(q.z, q.y, sh);
        t[2] = __builtin_amdgcn_alignbit(q.w, q.z, sh);
        t[3] = __builtin_amdgcn_alignbit(d4, q.w, sh);
        t[4] = d4 >> sh;
    } else {
        t[0] = q.x;
        t[1] = q.y;
        t[2] = q.z;
        t[3] = q.w;
        t[4] = d4;
    }
}

template <typename T> __device__ __forceinline__ void unpack_window(const uint32_t (&t)[5], typename T::C (&v)[16 / sizeof(typename T::S) + 1]) {
    using S = typename T::S;
    constexpr int E = 16 / sizeof(S);
    if constexpr (sizeof(S) == 2) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint16_t lo = static_cast<uint16_t>(t[i]), hi = static_cast<uint16_t>(t[i] >> 16);
            S a, b;
            __builtin_memcpy(&a, &lo, 2);
            __builtin_memcpy(&b, &hi, 2);
            if (2 * i <= E) v[2 * i] = widen<T>(a);
            if (2 * i + 1 <= E) v[2 * i + 1] = widen<T>(b);
        }
    } else {
#pragma unroll
        for (int i = 0; i <= E; ++i) {
            S a;
            __builtin_memcpy(&a, &t[i], 4);
            v[i] = widen<T>(a);
        }
    }
}

// NROWS staged rows (slots `slot0`, `slot0 + pitch`, ...) -> widened values of the thread's E + 1 columns
template <typename T, int NROWS>
__device__ __forceinline__ void read_rows(const char *slot0, int pitch, const RowRead &rr, const int *map, int ji,
                                          typename T::C (&v)[NROWS][16 / sizeof(typename T::S) + 1]) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S), E = 16 / ES;
    if (rr.affine) {
#pragma unroll
        for (int h = 0; h < NROWS; ++h) {
            uint32_t t[5];
            read_window<ES>(slot0 + h * pitch + rr.woff, rr.half, t);
            unpack_window<T>(t, v[h]);
        }
    } else {
        int cm[E + 1];
#pragma unroll
        for (int e = 0; e <= E; ++e) cm[e] = map[ji + e];
#pragma unroll
        for (int h = 0; h < NROWS; ++h) {
            const char *body = slot0 + h * pitch + rr.fboff;
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                const S raw = *reinterpret_cast<const S *>(body + (cm[e] > 0 ? cm[e] : 0) * ES);
                v[h][e] = cm[e] >= 0 ? widen<T>(raw) : CT(0);
            }
        }
    }
}

// the thread's E shifted columns of one staged row as raw elements (the sparse-shift input gradient: a copy)
template <typename T>
__device__ __forceinline__ Chunk<typename T::S, 16 / sizeof(typename T::S)> read_row_raw(const char *slot, const RowRead &rr,
                                                                                            const int *map, int ji) {
    using S = typename T::S;
    constexpr int ES = sizeof(S), E = 16 / ES;
    Chunk<S, E> c;
    if (rr.affine) {
        uint32_t t[5];
        read_window<ES>(slot + rr.woff, rr.half, t);
        __builtin_memcpy(c.e, t, 16);
    } else {
        S zero;
        __builtin_memset(&zero, 0, sizeof(S));
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int cm = map[ji + e];
            const S raw = *reinterpret_cast<const S *>(slot + rr.fboff + (cm > 0 ? cm : 0) * ES);
            c.e[e] = cm >= 0 ? raw : zero;
        }
    }
    return c;
}

// Column state of a lane: affine window (all valid columns consecutive and congruent with the workgroup's
// displacement) or element-wise reads.
template <int ES, int E>
__device__ __forceinline__ RowRead make_rowread(const int *map, int ji, bool live, int delta /*bytes, 0..15*/, int row_bytes) {
    RowRead rr;
    int base = 0;
    bool found = false, affine = true;
    int cm[E + 1];
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        cm[e] = live ? map[ji + e] : -1;
        if (!found && cm[e] >= 0) {
            base = cm[e] - e;
            found = true;
        }
    }
#pragma unroll
    for (int e = 0; e <= E; ++e) affine = affine && (cm[e] < 0 || cm[e] == base + e);
    const int disp = delta & 12;  // the rows of this kind sit `disp` bytes to the left of their slot body
    rr.half = (ES == 2 && (delta & 2)) ? 16 : 0;
    rr.fboff = kGuard - disp;
    int w = -kGuard;  // no valid column: any window inside the zeroed guard
    if (found) {
        const int b0 = base * ES;                    // source byte of column 0 of the window (may be negative)
        affine = affine && ((b0 & 15) == delta);     // congruent with the displacement (else: element-wise)
        w = (b0 & ~3) - disp;                        // a multiple of 16 when congruent
        w = w < -kGuard ? -kGuard : (w > row_bytes ? row_bytes : w);
    }
    rr.woff = kGuard + w;
    rr.affine = affine;
    return rr;
}

// One workgroup = one channel c and nseg segments; seg + 1 steps; see the header comment.
//   ND 2/3; ACTIVE: interpolating (active shift) or sparse shift; BACKWARD: grad_x + weight-gradient partials, else
//   the interpolating forward.  NP: 16-byte pieces a thread stages per step.
template <typename T, int ND, bool ACTIVE, bool BACKWARD, int NP>
__global__ __launch_bounds__(kThreads) void slide_kernel(const SlideParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S), E = 16 / ES;
    constexpr int NA = ND == 3 ? 2 : 1;
    constexpr int HALO = ND == 3 ? 1 : 0;
    constexpr int NDIFF = WDiff<ND>::N;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    static_assert(ACTIVE || BACKWARD, "the sparse-shift forward is served by the gather kernels");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double scratch[kThreads / 64];
    char *tiles = smem;
    int *maps = reinterpret_cast<int *>(smem + 2 * p.tile_bytes);
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2];
    const int *m0 = maps, *m1 = m0 + S0 + 1, *m2 = m1 + S1 + 1;
    int *gmaps = maps + S0 + S1 + S2 + 3;
    const int *g0 = gmaps, *g1 = g0 + S0 + 1, *g2 = g1 + S1 + 1;
    const int RB = S2 * ES, pitch = p.pitch;
    const int nseg = p.nseg;
    const int NSX = nseg + HALO;

    // ---- which channel / segments -------------------------------------------------------------------------------
    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int outer = fdiv(bid, p.d_inner), in = static_cast<int>(bid) - outer * p.inner;
    const int grp = fdiv(outer, p.d_C), c = outer - grp * p.C;
    int n0, a0 = 0, nhere, band3 = 0;
    if constexpr (ND == 3) {
        const int ag = fdiv(in, p.d_bands);
        band3 = in - ag * p.bands;
        n0 = grp;
        a0 = ag * nseg;
        nhere = min(nseg, S0 - a0);
    } else {
        n0 = fdiv(grp * nseg, p.d_bands);
        nhere = min(nseg, p.units - grp * nseg);
    }
    // segment idx of this workgroup -> element offset of its plane from (n0, c), first row, rows; false when absent
    auto seg_info = [&](int idx, int &poff, int &bstart, int &len) {
        if constexpr (ND == 3) {
            poff = 0;  // the a-plane is added by the caller (it goes through the plane maps)
            bstart = band3 * p.seg;
            len = min(p.seg, S1 - bstart);
            return idx < nhere;
        } else {
            const int q = grp * nseg + idx;
            const int n = fdiv(q, p.d_bands), band = q - n * p.bands;
            poff = (n - n0) * p.C * static_cast<int>(p.plane);
            bstart = band * p.seg;
            len = min(p.seg, S1 - bstart);
            return idx < nhere;
        }
    };

    // ---- per-channel shift, maps, zeroed tiles ------------------------------------------------------------------
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.wcol[d] >= 0) {
            const CT wv = load_weight<CT>(p.w, p.wkind, c * p.nd + p.wcol[d]);
            if constexpr (BACKWARD) prep_shift_backward<CT>(wv, ACTIVE, sh[d], dw[p.wcol[d]]);
            else prep_shift_forward<CT>(wv, true, sh[d], dw[p.wcol[d]]);
        }
    }
    build_maps(maps, p.S, sh, -1, p.pad);
    // grad_x source: the sparse shift reads grad_out at o + shift, the active one at o - shift (shifts_kernels.h:287-293)
    if constexpr (BACKWARD) build_maps(gmaps, p.S, sh, ACTIVE ? -1 : +1, p.pad);
    {
        u4 *z = reinterpret_cast<u4 *>(tiles);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = threadIdx.x; i < 2 * p.tile_bytes / 16; i += kThreads) z[i] = zero;
    }
    // byte phase (mod 16) of the affine part of the column maps: rows are staged that many bytes (rounded down to a
    // dword) to the left, which puts every thread's window on a 16-byte boundary
    const int csx = canon_shift(sh[2], S2, p.pad, p.d_per);
    const int dx = (-csx * ES) & 15;
    const int dg = (ACTIVE || !BACKWARD) ? dx : ((-canon_shift(-sh[2], S2, p.pad, p.d_per) * ES) & 15);
    __syncthreads();

    // ---- this thread's chunk column ------------------------------------------------------------------------------
    const int s = fdiv(threadIdx.x, p.d_cpr), tc = static_cast<int>(threadIdx.x) - s * p.cpr;
    int mypoff, mybstart, mylen;
    const bool worker = s < nseg && seg_info(s, mypoff, mybstart, mylen);
    if (!worker) mylen = -1;
    const int ji = tc * E;
    const RowRead rx = make_rowread<ES, E>(m2, ji, worker, dx, RB);
    const RowRead rg = BACKWARD ? make_rowread<ES, E>(g2, ji, worker, dg, RB) : rx;
    const int oX = s * pitch;                                  // slot bases (bytes from the tile start)
    const int oG = (NSX + s) * pitch + kGuard + tc * 16;        // the incoming gradient at the thread's own position
    const int oGS = (NSX + nseg + s) * pitch;
    const int64_t plane_nc = static_cast<int64_t>(n0) * p.C + c;
    const S *xb = static_cast<const S *>(p.x) + plane_nc * p.plane;
    const S *gb = BACKWARD ? static_cast<const S *>(p.go) + plane_nc * p.plane : xb;
    S *outp = static_cast<S *>(p.out) + plane_nc * p.plane + (worker ? mypoff + (ND == 3 ? (a0 + s) * S1 * S2 : 0) + ji : 0);

    // ---- the pieces this thread stages every step -----------------------------------------------------------------
    // kind 0: x rows (through the x maps), 1: grad_out rows at the output position, 2: grad_out rows through the grad maps
    int pbase[NP], pmeta[NP], pdst[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        pbase[k] = -1;
        pmeta[k] = 0;
        pdst[k] = -1;
        if (q < p.npieces) {
            const int slot = fdiv(q, p.d_cpr), j = q - slot * p.cpr;
            const int kind = slot < NSX ? 0 : (slot < NSX + nseg ? 1 : 2);
            const int idx = slot - (kind == 0 ? 0 : (kind == 1 ? NSX : NSX + nseg));
            int poff, bstart, len;
            bool ok;
            if constexpr (ND == 3) {
                seg_info(0, poff, bstart, len);
                const int a = a0 + idx;
                int pa = -1;
                if (kind == 0) pa = idx <= nhere ? m0[a] : -1;
                else if (kind == 1) pa = idx < nhere ? a : -1;
                else if (ACTIVE) pa = idx <= nhere ? g0[a] : -1;
                else pa = idx < nhere ? g0[a] : -1;
                ok = pa >= 0;
                poff = pa * S1 * S2;
            } else {
                ok = seg_info(idx, poff, bstart, len);
            }
            pbase[k] = ok ? poff + j * E : -1;
            pmeta[k] = kind | (bstart << 2) | (len << 17);
            const int disp = kind == 0 ? (dx & 12) : (kind == 2 ? (dg & 12) : 0);
            pdst[k] = slot * pitch + kGuard + j * 16 - disp;
        }
    }
    u4 pv[NP];
    auto issue_loads = [&](int t1) {  // the rows step t1 needs, into registers
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int kind = pmeta[k] & 3, bstart = (pmeta[k] >> 2) & 0x7fff, len = pmeta[k] >> 17;
            int row = -1;
            if (kind == 0) {
                if (t1 <= len) row = m1[bstart + t1];
            } else if (kind == 1) {
                if (t1 >= 1 && t1 <= len) row = bstart + t1 - 1;
            } else if (ACTIVE) {
                if (t1 <= len) row = g1[bstart + t1];
            } else {
                if (t1 >= 1 && t1 <= len) row = g1[bstart + t1 - 1];
            }
            const u4 zero = {0u, 0u, 0u, 0u};
            pv[k] = zero;
            if (pbase[k] >= 0 && row >= 0) {
                const S *src = (kind == 0 ? xb : gb) + (static_cast<int64_t>(pbase[k]) + static_cast<int64_t>(row) * S2);
                pv[k] = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(src, 16));
            }
        }
    };
    auto write_tile = [&](char *tile) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (pdst[k] >= 0) {
                uint32_t *d = reinterpret_cast<uint32_t *>(tile + pdst[k]);  // dword aligned (displaced rows)
                d[0] = pv[k].x;
                d[1] = pv[k].y;
                d[2] = pv[k].z;
                d[3] = pv[k].w;
            }
        }
    };

    // ---- walk ---------------------------------------------------------------------------------------------------
    const CT dA = dw[0], dB = dw[ND - 2], dI = dw[ND - 1];
    CT xp[NA][E + 1];   // x rows of the previous step (the hb = 0 corners)
    CT lp[E + 1];       // ACTIVE: grad_out (forward: x) row of the previous step, blended over dim0 in 3-D
    CT part[NDIFF];
    double dsum[NDIFF];
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) {
        part[i] = CT(0);
        dsum[i] = 0.0;
    }
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int e = 0; e <= E; ++e) xp[h][e] = CT(0);
#pragma unroll
    for (int e = 0; e <= E; ++e) lp[e] = CT(0);

    issue_loads(0);
    for (int t = 0; t <= p.seg; ++t) {
        char *tile = tiles + (t & 1) * p.tile_bytes;
        write_tile(tile);
        __syncthreads();
        if (t < p.seg) issue_loads(t + 1);  // in flight while this step is computed
        if (t <= mylen) {
            if constexpr (BACKWARD) {
                CT xn[NA][E + 1];
                read_rows<T, NA>(tile + oX, pitch, rx, m2, ji, xn);
                if (t >= 1) {
                    // weight-gradient sums: g * corner differences (corner_diffs, shiftnd_common.hpp) with the
                    // differences that neighbouring elements / rows share computed once
                    Chunk<S, E> gch;
                    __builtin_memcpy(gch.e, __builtin_assume_aligned(tile + oG, 16), 16);
                    if constexpr (ND == 3) {
                        CT P[E + 1], Q[E + 1];
#pragma unroll
                        for (int e = 0; e <= E; ++e) {
                            P[e] = xn[0][e] - xp[0][e];
                            Q[e] = xn[1][e] - xp[1][e];
                        }
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const CT gval = widen<T>(gch.e[e]);
                            part[0] = fma_ct(gval, P[e], part[0]);
                            part[1] = fma_ct(gval, Q[e], part[1]);
                            part[2] = fma_ct(gval, P[e + 1], part[2]);
                            part[3] = fma_ct(gval, Q[e + 1], part[3]);
                            part[4] = fma_ct(gval, xp[0][e + 1] - xp[0][e], part[4]);
                            part[5] = fma_ct(gval, xp[1][e + 1] - xp[1][e], part[5]);
                            part[6] = fma_ct(gval, xn[0][e + 1] - xn[0][e], part[6]);
                            part[7] = fma_ct(gval, xn[1][e + 1] - xn[1][e], part[7]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const CT gval = widen<T>(gch.e[e]);
                            part[0] = fma_ct(gval, xp[0][e + 1] - xp[0][e], part[0]);
                            part[1] = fma_ct(gval, xn[0][e + 1] - xn[0][e], part[1]);
                        }
                    }
                    if ((t & (kFlushSteps - 1)) == 0) {
#pragma unroll
                        for (int i = 0; i < NDIFF; ++i) {
                            dsum[i] += static_cast<double>(part[i]);
                            part[i] = CT(0);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < NA; ++h)
#pragma unroll
                    for (int e = 0; e <= E; ++e) xp[h][e] = xn[h][e];
            }
            S *dst = outp + static_cast<int64_t>(mybstart + t - 1) * S2;
            if constexpr (ACTIVE) {
                CT gn[NA][E + 1];
                if constexpr (BACKWARD) read_rows<T, NA>(tile + oGS, pitch, rg, g2, ji, gn);
                else read_rows<T, NA>(tile + oX, pitch, rx, m2, ji, gn);
                CT ln[E + 1];
#pragma unroll
                for (int e = 0; e <= E; ++e) {
                    if constexpr (ND == 3) ln[e] = lerp_t<T>(gn[0][e], gn[1][e], dA);
                    else ln[e] = gn[0][e];
                }
                if (t >= 1) {
                    CT m[E + 1];
#pragma unroll
                    for (int e = 0; e <= E; ++e) m[e] = lerp_t<T>(lp[e], ln[e], dB);
                    Chunk<S, E> res;
#pragma unroll
                    for (int e = 0; e < E; ++e) res.e[e] = narrow<T>(lerp_t<T>(m[e], m[e + 1], dI));
                    store_chunk<S, E>(dst, res);
                }
#pragma unroll
                for (int e = 0; e <= E; ++e) lp[e] = ln[e];
            } else {
                if (t >= 1) store_chunk<S, E>(dst, read_row_raw<T>(tile + oGS, rg, g2, ji));
            }
        }
    }

    if constexpr (BACKWARD) {
#pragma unroll
        for (int i = 0; i < NDIFF; ++i) dsum[i] += static_cast<double>(part[i]);
        double acc[3] = {0.0, 0.0, 0.0};
        const double dwd[3] = {static_cast<double>(dw[0]), static_cast<double>(dw[1]), static_cast<double>(dw[2])};
        blend_diffs<ND>(dsum, dwd, acc);
        const int pidx = grp * p.inner + in;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tsum = block_sum(acc[k], scratch);
            if (threadIdx.x == 0) p.partials[(static_cast<size_t>(pidx) * p.C + c) * 3 + k] = tsum;
        }
    }
}

// =====================================================================================================
// Host side
// =====================================================================================================
// diagnostics (shiftnd_set_tuning knobs 12..15): 12 = which problems take these kernels (bit 0: 3-D, bit 1: 2-D;
// default set in slide_wanted), 13 = workgroups wanted (0 = automatic), 14 = minimum rows per band
thread_local int g_slide_tune[4] = {-1, 0, 16, 0};

struct SlidePlan {
    int cpr, nseg, seg, bands, agroups, inner, units, groups, pitch, tile_bytes, npieces, nslots;
    size_t lds;
    unsigned grid;
    bool ok;
};

bool contiguous5(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

SlidePlan slide_plan(const Geometry &g, int es, bool backward, int np_max) {
    SlidePlan pl{};
    pl.ok = false;
    if (g.nd != 2 && g.nd != 3) return pl;
    if (es != 2 && es != 4) return pl;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t RB = g.S[2] * es;
    if (RB % 16 != 0 || RB / 16 > kThreads || g.S[1] > 16383 || g.S[0] > 16383) return pl;
    const int64_t plane = g.S[0] * g.S[1] * g.S[2];
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return pl;
    pl.cpr = static_cast<int>(RB / 16);
    const int halo = g.nd == 3 ? 1 : 0;
    const int kinds = backward ? 3 : 1;
    // segments: as many as fit 256 threads and np_max pieces per thread
    int nseg = kThreads / pl.cpr;
    if (g.nd == 3 && nseg > g.S[0]) nseg = static_cast<int>(g.S[0]);
    auto slots_of = [&](int n) { return backward ? (n + halo) + n + (g.active ? n + halo : n) : n + halo; };
    const size_t map_bytes = static_cast<size_t>(g.S[0] + g.S[1] + g.S[2] + 3) * (backward ? 2 : 1) * sizeof(int);
    auto lds_of = [&](int n) { return 2 * static_cast<size_t>(slots_of(n)) * (RB + 2 * kGuard) + map_bytes; };
    while (nseg > 1 && (static_cast<int64_t>(slots_of(nseg)) * pl.cpr > static_cast<int64_t>(np_max) * kThreads ||
                        lds_of(nseg) > 64 * 1024))
        --nseg;
    if (static_cast<int64_t>(slots_of(nseg)) * pl.cpr > static_cast<int64_t>(np_max) * kThreads) return pl;
    (void)kinds;
    const int64_t min_wgs = g_slide_tune[1] > 0 ? g_slide_tune[1] : 4096;
    const int64_t min_rows = g_slide_tune[2] > 0 ? g_slide_tune[2] : 16;
    if (g.nd == 3) {
        if (plane >= (1LL << 30)) return pl;
        pl.agroups = static_cast<int>((g.S[0] + nseg - 1) / nseg);
        const int64_t base = g.N * g.C * pl.agroups;
        int64_t bands = (min_wgs + base - 1) / base;
        const int64_t max_bands = g.S[1] / min_rows > 0 ? g.S[1] / min_rows : 1;
        if (bands > max_bands) bands = max_bands;
        if (bands < 1) bands = 1;
        pl.seg = static_cast<int>((g.S[1] + bands - 1) / bands);
        pl.bands = static_cast<int>((g.S[1] + pl.seg - 1) / pl.seg);
        pl.inner = pl.agroups * pl.bands;
        pl.units = 0;
        pl.groups = static_cast<int>(g.N);
    } else {
        // units = (n, band); a workgroup takes nseg consecutive units of one channel
        int64_t bands = 1;
        const int64_t max_bands = g.S[1] / min_rows > 0 ? g.S[1] / min_rows : 1;
        while (bands < max_bands && g.C * ((g.N * bands + nseg - 1) / nseg) < min_wgs) ++bands;
        pl.seg = static_cast<int>((g.S[1] + bands - 1) / bands);
        pl.bands = static_cast<int>((g.S[1] + pl.seg - 1) / pl.seg);
        pl.agroups = 1;
        pl.inner = 1;
        const int64_t units = g.N * pl.bands;
        if (units >= (1LL << 30)) return pl;
        pl.units = static_cast<int>(units);
        pl.groups = static_cast<int>((units + nseg - 1) / nseg);
        // the planes of one workgroup are addressed with 32-bit element offsets from its first plane
        const int64_t span = (static_cast<int64_t>(nseg) / pl.bands + 2) * g.C * plane;
        if (span >= (1LL << 30)) return pl;
    }
    pl.nseg = nseg;
    pl.nslots = slots_of(nseg);
    pl.npieces = pl.nslots * pl.cpr;
    pl.pitch = static_cast<int>(RB) + 2 * kGuard;
    pl.tile_bytes = pl.nslots * pl.pitch;
    pl.lds = 2 * static_cast<size_t>(pl.tile_bytes) + map_bytes;
    if (pl.lds > 64 * 1024) return pl;
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C * pl.inner;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

void fill_slide(SlideParams &p, const Geometry &g, const SlidePlan &pl) {
    p.plane = g.S[0] * g.S[1] * g.S[2];
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
    }
    p.cpr = pl.cpr;
    p.nseg = pl.nseg;
    p.seg = pl.seg;
    p.bands = pl.bands;
    p.agroups = pl.agroups;
    p.inner = pl.inner;
    p.units = pl.units;
    p.pitch = pl.pitch;
    p.tile_bytes = pl.tile_bytes;
    p.npieces = pl.npieces;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(pl.cpr));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(pl.bands));
    p.d_inner = make_fastdiv(static_cast<uint32_t>(pl.inner));
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    p.d_per = make_fastdiv(static_cast<uint32_t>(map_period(static_cast<int>(g.S[2]), g.pad)));
}

constexpr int kNpBackward = 3, kNpForward = 2;

template <typename T, bool ACTIVE>
void launch_slide_backward(const SlideParams &p, const SlidePlan &pl, hipStream_t st) {
    if (p.nd == 3) hipLaunchKernelGGL((slide_kernel<T, 3, ACTIVE, true, kNpBackward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    else hipLaunchKernelGGL((slide_kernel<T, 2, ACTIVE, true, kNpBackward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
}
template <typename T>
void launch_slide_forward(const SlideParams &p, const SlidePlan &pl, hipStream_t st) {
    if (p.nd == 3) hipLaunchKernelGGL((slide_kernel<T, 3, true, false, kNpForward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    else hipLaunchKernelGGL((slide_kernel<T, 2, true, false, kNpForward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
}

bool slide_wanted(const Geometry &g) {
    const int mode = g_slide_tune[0] >= 0 ? g_slide_tune[0] : 1;  // default: 3-D problems
    return (g.nd == 3 && (mode & 1)) || (g.nd == 2 && (mode & 2));
}

}  // namespace

void slide_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 4) g_slide_tune[knob] = value;
}

bool slide_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) return false;
    if (!slide_wanted(g)) return false;
    if (!contiguous5(g.xs, g.N, g.C, g.S) || !contiguous5(g.os, g.N, g.C, g.O) || !contiguous5(g.gs, g.N, g.C, g.S)) return false;
    if (reinterpret_cast<uintptr_t>(go) % 16 || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(gx) % 16) return false;
    return slide_plan(g, dtype_size(dtype), true, kNpBackward).ok;
}

size_t slide_backward_workspace(const Geometry &g, int dtype) {
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) return 0;
    const SlidePlan pl = slide_plan(g, dtype_size(dtype), true, kNpBackward);
    if (!pl.ok) return 0;
    return static_cast<size_t>(pl.groups) * pl.inner * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

int slide_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st) {
    const SlidePlan pl = slide_plan(g, dtype_size(dtype), true, kNpBackward);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    SlideParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    fill_slide(p, g, pl);
    note_kernel("slide_backward");
    const int cn = p.C * p.nd, groups = pl.groups * pl.inner;
#define SHIFTND_SLIDE_BWD(TT) \
    { \
        if (g.active) launch_slide_backward<TT, true>(p, pl, st); \
        else launch_slide_backward<TT, false>(p, pl, st); \
        hipLaunchKernelGGL((reduce_weight_grads<TT>), dim3(cn), dim3(64), 0, st, p.partials, groups, p.C, p.nd, \
                           static_cast<typename TT::S *>(gw)); \
    }
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_SLIDE_BWD(f32_t) break;
    case SHIFTND_F16: SHIFTND_SLIDE_BWD(f16_t) break;
    default: SHIFTND_SLIDE_BWD(bf16_t) break;
    }
#undef SHIFTND_SLIDE_BWD
    return SHIFTND_OK;
}

bool slide_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) return false;
    if (!g.active || !slide_wanted(g)) return false;
    if (!contiguous5(g.xs, g.N, g.C, g.S) || !contiguous5(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16) return false;
    return slide_plan(g, dtype_size(dtype), false, kNpForward).ok;
}

int slide_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st) {
    const SlidePlan pl = slide_plan(g, dtype_size(dtype), false, kNpForward);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    SlideParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = dtype;
    fill_slide(p, g, pl);
    note_kernel("slide_forward");
    switch (dtype) {
    case SHIFTND_F32: launch_slide_forward<f32_t>(p, pl, st); break;
    case SHIFTND_F16: launch_slide_forward<f16_t>(p, pl, st); break;
    default: launch_slide_forward<bf16_t>(p, pl, st); break;
    }
    return SHIFTND_OK;
}

}  // namespace shiftnd
