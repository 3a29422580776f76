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
//   * rows are staged exactly as they lie in memory (aligned 16-byte pieces behind a zeroed guard).  A thread's window
//     of E + 1 shifted columns starts `phase` bytes into an aligned 32-byte span of its row; the phase is the same for
//     every thread of the workgroup (one channel = one inner shift), so it is resolved by a uniform switch into
//     compile-time register naming (RowRead / read_windows): two aligned ds_read_b128 per row, one v_alignbit per dword
//     only for the odd half of 16-bit data.  Every row has zeroed guard bytes on both sides, so with zeros padding the
//     columns outside the row read as 0 without a mask; chunks whose column map is not affine, or not congruent with
//     the workgroup's phase (edges of the wrapping / clamping paddings), read element by element through the LDS
//     column map.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/): backward kernels/shifts_kernels.h:222-327 with
// kernels/interpolation.h:3-61, forward kernels/shifts_kernels.h:156-220; weight preparation cpu/shifts_cpu.cpp:223-224,
// :242-244.  Roofline: HBM; backward 3*s bytes per element, forward 2*s.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kGuard = 32;          // zeroed bytes in front of every staged row (and behind the last one)
constexpr int kFlushSteps = 8;      // fp32 partial sums are blended into the fp64 accumulators every kFlushSteps rows

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
    int zslot;           // byte offset, in a tile, of a slot that is never written (all zeros): stands in for fill rows
    int npieces;         // 16-byte pieces staged per step
    unsigned xcd_blocks;
    FastDiv d_cpr, d_bands, d_inner, d_C;
    FastDiv d_per[3];    // divide by the padding period of each dim
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

// acc + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): the products of two
// 16-bit values are exact in fp32, and nothing has to be widened first (on gfx950 a widening shift costs as much issue
// time as this whole instruction; tools/valu_bench.hip)
template <typename T> __device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float c) {
    if constexpr (T::kDtype == SHIFTND_BF16) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        v2 x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        return __builtin_amdgcn_fdot2_f32_bf16(x, y, c, false);
    } else {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        v2 x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        return __builtin_amdgcn_fdot2(x, y, c, false);
    }
}

// How a thread reads its E + 1 shifted columns of a staged row (one per tensor kind: x rows, grad_out rows).
// Rows are staged as they are in memory (16-byte pieces at 16-byte LDS addresses: conflict-free ds_write_b128), so a
// thread's window of E + 1 columns starts `phase` bytes into an aligned 32-byte span; `phase` (0..15) is the same for
// every thread of the workgroup (one channel = one shift): dword R = phase / 4, and for 16-bit data the odd half.
struct RowRead {
    int woff;     // affine lanes: byte offset, from the row's slot base, of the aligned 32-byte span holding the window
    int phase;    // uniform: byte phase of the window inside the span (see above)
    bool affine;
};

// E + 1 raw elements as 5 dwords (first element in the low bits of t[0]) out of the aligned 32-byte span at p.
// R / HALF are compile-time: which of the 8 dwords the window starts with is register naming, and the funnel shift
// for the odd half disappears when HALF == 0.
template <int ES, int R, int HALF> __device__ __forceinline__ void read_window_ct(const char *p, uint32_t (&t)[5]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q0 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(p, 16));
    const u4 q1 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(p + 16, 16));
    const uint32_t d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    if constexpr (ES == 2 && HALF != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_alignbit(d[R + i + 1], d[R + i], 16);
        t[4] = d[R + 4] >> 16;
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) t[i] = d[R + i];
    }
}
template <int ES, int NROWS>
__device__ __forceinline__ void read_windows(const char *const (&slot)[NROWS], int woff, int phase, uint32_t (&t)[NROWS][5]) {
    // uniform switch: one workgroup = one phase
#define SHIFTND_WIN_CASE(RR, HH) \
    case (RR * 2 + HH): \
        _Pragma("unroll") for (int h = 0; h < NROWS; ++h) read_window_ct<ES, RR, HH>(slot[h] + woff, t[h]); \
        break;
    switch (__builtin_amdgcn_readfirstlane(ES == 2 ? (phase >> 1) : ((phase >> 2) << 1))) {
        SHIFTND_WIN_CASE(0, 0)
        SHIFTND_WIN_CASE(1, 0)
        SHIFTND_WIN_CASE(2, 0)
        SHIFTND_WIN_CASE(3, 0)
        SHIFTND_WIN_CASE(0, 1)
        SHIFTND_WIN_CASE(1, 1)
        SHIFTND_WIN_CASE(2, 1)
    default:
        _Pragma("unroll") for (int h = 0; h < NROWS; ++h) read_window_ct<ES, 3, 1>(slot[h] + woff, t[h]);
        break;
    }
#undef SHIFTND_WIN_CASE
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

// The thread's E + 1 shifted columns of NROWS staged rows (`slot[h]`: first byte of the row's slot) as packed windows:
// 5 dwords per row, first column in the low bits of t[.][0] (16-bit types: two columns per dword).
template <typename T, int NROWS>
__device__ __forceinline__ void read_rows(const char *const (&slot)[NROWS], const RowRead &rr, const int *map, int ji,
                                          uint32_t (&t)[NROWS][5]) {
    using S = typename T::S;
    constexpr int ES = sizeof(S), E = 16 / ES;
    if (rr.affine) {
        read_windows<ES, NROWS>(slot, rr.woff, rr.phase, t);
    } else {
        int cm[E + 1];
#pragma unroll
        for (int e = 0; e <= E; ++e) cm[e] = map[ji + e];
#pragma unroll
        for (int h = 0; h < NROWS; ++h) {
            const char *body = slot[h] + kGuard;
            uint32_t raw[E + 1];
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                typename raw_t<ES>::type r = *reinterpret_cast<const typename raw_t<ES>::type *>(body + (cm[e] > 0 ? cm[e] : 0) * ES);
                raw[e] = cm[e] >= 0 ? static_cast<uint32_t>(r) : 0u;
            }
            if constexpr (ES == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) t[h][i] = raw[2 * i] | (raw[2 * i + 1] << 16);
                t[h][4] = raw[8];
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) t[h][i] = raw[i];
            }
        }
    }
}

// Column state of a lane: affine window (all valid columns consecutive and congruent with the workgroup's
// displacement) or element-wise reads.
template <int ES, int E>
__device__ __forceinline__ RowRead make_rowread(const int *map, int ji, bool live, int delta /*bytes, 0..15*/, int row_bytes) {
    (void)row_bytes;
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
    rr.phase = delta;
    int w = -kGuard;  // no valid column: a span inside the zeroed guard
    if (found) {
        const int b0 = base * ES;                    // source byte of column 0 of the window (may be negative)
        affine = affine && ((b0 & 15) == delta);     // congruent with the workgroup's phase (else: element-wise)
        w = b0 & ~15;                                // >= -16 (base >= -E); the 32-byte span ends at most 16 bytes behind the
                                                     // row, inside the guard: b0 <= row_bytes - ES, so w <= row_bytes - 16
        w = w < -kGuard ? -kGuard : w;
    }
    rr.woff = kGuard + w;
    rr.affine = affine;
    return rr;
}

// One workgroup = one channel c and nseg segments; seg + 1 steps; see the header comment.
//   ND 2/3; ACTIVE: interpolating (active shift) or sparse shift; BACKWARD: grad_x + weight-gradient partials, else
//   the interpolating forward.  NP: 16-byte pieces a thread stages per step.
template <typename T, int ND, bool ACTIVE, bool BACKWARD, int NP, int DEPTH>
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
    {
        u4 *z = reinterpret_cast<u4 *>(tiles);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = threadIdx.x; i < 2 * p.tile_bytes / 16; i += kThreads) z[i] = zero;
    }
    // byte phase (mod 16) of the affine part of the column maps: where, inside an aligned 32-byte span of a staged
    // row, a thread's window of shifted columns starts (RowRead)
    const int csx = canon_shift(sh[2], S2, p.pad, p.d_per[2]);
    const int dx = (-csx * ES) & 15;
    const int dg = (ACTIVE || !BACKWARD) ? dx : ((-canon_shift(-sh[2], S2, p.pad, p.d_per[2]) * ES) & 15);
    __syncthreads();

    // ---- this thread's chunk column ------------------------------------------------------------------------------
    const int s = fdiv(threadIdx.x, p.d_cpr), tc = static_cast<int>(threadIdx.x) - s * p.cpr;
    int mypoff, mybstart, mylen;
    const bool worker = s < nseg && seg_info(s, mypoff, mybstart, mylen);
    if (!worker) mylen = -1;
    const int ji = tc * E;
    const RowRead rx = make_rowread<ES, E>(m2, ji, worker, dx, RB);
    const RowRead rg = BACKWARD ? make_rowread<ES, E>(g2, ji, worker, dg, RB) : rx;
    // planes this thread's rows come from: bit h = x plane (a + h) is not fill, bit 2 + h = grad plane likewise
    int pvalid = 0xf;
    if constexpr (ND == 3) {
        if (worker) {
            pvalid = (m0[a0 + s] >= 0 ? 1 : 0) | (m0[a0 + s + 1] >= 0 ? 2 : 0);
            if constexpr (BACKWARD) pvalid |= (g0[a0 + s] >= 0 ? 4 : 0) | ((ACTIVE ? g0[a0 + s + 1] : 0) >= 0 ? 8 : 0);
        }
    }
    const int oX = s * pitch;                                  // slot bases (bytes from the tile start)
    const int oG = (NSX + s) * pitch + kGuard + tc * 16;        // the incoming gradient at the thread's own position
    const int oGS = (NSX + nseg + s) * pitch;
    const int64_t plane_nc = static_cast<int64_t>(n0) * p.C + c;
    const S *xb = static_cast<const S *>(p.x) + plane_nc * p.plane;
    const S *gb = BACKWARD ? static_cast<const S *>(p.go) + plane_nc * p.plane : xb;

    // ---- the pieces this thread stages every step -----------------------------------------------------------------
    // kind 0: x rows (through the x maps), 1: grad_out rows at the output position, 2: grad_out rows through the grad maps
    // pdst: LDS byte offset of the piece in a tile | kind << 16; pbase: element offset of the piece's plane and
    // column from (n0, c); pseg (2-D only; uniform in 3-D): first row | rows << 15 of the piece's segment
    int pbase[NP], pdst[NP], pseg[ND == 2 ? NP : 1];
    int ubstart = 0, ulen = 0;  // 3-D: every segment of the workgroup covers the same rows
    if constexpr (ND == 3) {
        int upoff;
        seg_info(0, upoff, ubstart, ulen);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        pbase[k] = 0;
        pdst[k] = p.zslot + pitch + kGuard;  // no k-th piece: loads an x row like everybody, stores it in the dump slot (a
                                    // store under a thread-dependent branch would make hipcc wait for ALL memory traffic)
        if constexpr (ND == 2) pseg[k] = 0;
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
            pbase[k] = ok ? poff + j * E : 0;  // (absent planes: any readable address; their slots are never read)
            if constexpr (ND == 2) pseg[k] = bstart | (len << 15);
            pdst[k] = (slot * pitch + kGuard + j * 16) | (kind << 16);
        }
    }
    u4 pv[DEPTH][NP];  // staged pieces in flight: DEPTH steps ahead
    // which row of its plane a piece of kind `kind` loads for step t1 (-1: fill, or nothing needed)
    auto row_of = [&](int kind, int bstart, int len, int t1) {
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
        return row;
    };
    auto issue_loads = [&](int t1, u4 (&pvr)[NP]) {  // the rows step t1 needs, into registers
        int urow[3] = {-1, -1, -1};
        if constexpr (ND == 3) {  // one row per kind for the whole workgroup: scalar
#pragma unroll
            for (int kind = 0; kind < (BACKWARD ? 3 : 1); ++kind)
                urow[kind] = __builtin_amdgcn_readfirstlane(row_of(kind, ubstart, ulen, t1));
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int kind = (pdst[k] >> 16) & 3;
            int row;
            if constexpr (ND == 3) row = kind == 0 ? urow[0] : (kind == 1 ? urow[1] : urow[2]);
            else row = row_of(kind, pseg[k] & 0x7fff, pseg[k] >> 15, t1);
            // unconditional (a conditional load would merge with a constant and be waited for right here): rows that
            // are fill or not needed load row 0 -- the readers take the all-zero slot instead, see row_slot()
            row = row < 0 ? 0 : row;
            const S *src = (kind == 0 ? xb : gb) + (static_cast<int64_t>(pbase[k]) + static_cast<int64_t>(row) * S2);
            pvr[k] = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(src, 16));
        }
    };
    auto write_tile = [&](char *tile, const u4 (&pvr)[NP]) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            *reinterpret_cast<u4 *>(__builtin_assume_aligned(tile + (pdst[k] & 0xffff), 16)) = pvr[k];
        }
    };

    // ---- walk ---------------------------------------------------------------------------------------------------
    const CT dA = dw[0], dB = dw[ND - 2], dI = dw[ND - 1];
    uint32_t xpw[NA][5];  // x rows of the previous step (the hb = 0 corners), as packed windows
    CT lp[E + 1];         // ACTIVE: grad_out (forward: x) row of the previous step, blended over dim0 in 3-D
    // Weight-gradient sums over the last <= kFlushSteps rows, in the compute type.
    //   4-byte data: part[i] = sum g * corner difference i (corner_diffs, shiftnd_common.hpp)
    //   2-byte data: per staged x row r (previous rows first, then the new ones) A_r = sum_e g(e) x_r(e) in part[2r] and
    //   B_r = sum_e g(e) x_r(e + 1) in part[2r + 1], from the PACKED rows with dot2 instructions; every corner
    //   difference sum is a difference of two of them, taken in fp64 when the block is flushed (the tolerance for the
    //   weight gradient of 16-bit tensors is the 16-bit epsilon; measured error ~1e-6 relative)
    constexpr int NPART = ES == 2 ? 4 * NA : NDIFF;
    CT part[NPART];
    double acc[3] = {0.0, 0.0, 0.0};  // the thread's weight-gradient partials: blends of the flushed sums, in fp64
#pragma unroll
    for (int i = 0; i < NPART; ++i) part[i] = CT(0);
    auto flush = [&]() {  // blend_diffs is linear in the sums: blending each flushed block equals blending the total
        const double dwd[3] = {static_cast<double>(dw[0]), static_cast<double>(dw[1]), static_cast<double>(dw[2])};
        double pd[NPART], sd[NDIFF], gb[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NPART; ++i) {
            pd[i] = static_cast<double>(part[i]);
            part[i] = CT(0);
        }
        if constexpr (ES == 2 && ND == 3) {
            // rows: 0 = (prev, ha0), 1 = (prev, ha1), 2 = (new, ha0), 3 = (new, ha1); A = pd[2r], B = pd[2r + 1]
            sd[0] = pd[4] - pd[0];
            sd[1] = pd[6] - pd[2];
            sd[2] = pd[5] - pd[1];
            sd[3] = pd[7] - pd[3];
            sd[4] = pd[1] - pd[0];
            sd[5] = pd[3] - pd[2];
            sd[6] = pd[5] - pd[4];
            sd[7] = pd[7] - pd[6];
        } else if constexpr (ES == 2) {
            sd[0] = pd[1] - pd[0];
            sd[1] = pd[3] - pd[2];
        } else {
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) sd[i] = pd[i];
        }
        blend_diffs<ND>(sd, dwd, gb);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += gb[k];
    };
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int i = 0; i < 5; ++i) xpw[h][i] = 0u;
#pragma unroll
    for (int e = 0; e <= E; ++e) lp[e] = CT(0);

    Chunk<S, E> pres{};  // output chunk computed by the previous step, and its row (-1: none)
    int prow = -1;
    // The store is unconditional (a store under a thread-dependent branch lowers the number of operations hipcc can
    // count on between a load and its use, and the prefetch is waited for too early): threads with nothing to store use a
    // buffer offset beyond the resource, which the hardware drops.
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<S *>(p.out) + plane_nc * p.plane, 0, 0x7ffffffc, 0x00020000);   // spans below 2^31 bytes (slide_plan)
    const uint32_t obase = static_cast<uint32_t>(worker ? mypoff + (ND == 3 ? (a0 + s) * S1 * S2 : 0) + ji : 0) * ES;
    auto flush_store = [&]() {
        // a plain (L2-retained) store: a step writes one 16-byte-multiple row segment per plane, and the rows of a
        // plane follow each other a step apart -- L2 merges them into whole lines; as nontemporal stores the partial
        // lines went to HBM one by one (C3: forward 0.222 -> 0.184 ms, backward 0.324 -> 0.287 ms)
        u4 data;
        __builtin_memcpy(&data, pres.e, 16);
        const uint32_t off = prow >= 0 ? obase + static_cast<uint32_t>(prow) * static_cast<uint32_t>(S2 * ES) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(data, ores, off, 0, 0);
        prow = -1;
    };
    auto step = [&](int t, u4 (&pvr)[NP]) {
        char *tile = tiles + (t & 1) * p.tile_bytes;
        write_tile(tile, pvr);
        __syncthreads();
        // The output row of the previous step leaves here, BEFORE the next loads are issued: loads and stores share
        // one in-order counter (vmcnt), and the compiler waits for everything older when it needs the loads back, so a
        // store issued after the loads would be waited for (its whole write latency) at the top of every step.
        flush_store();
        issue_loads(t + DEPTH, pvr);  // in flight while this and the next DEPTH - 1 steps are computed (beyond the band: row 0, unused)
        if (t <= mylen) {
            // fill rows (zeros padding) read the all-zero slot
            const bool rvx = m1[(ND == 3 ? ubstart : mybstart) + t] >= 0;
            const char *xs[NA];
#pragma unroll
            for (int h = 0; h < NA; ++h) xs[h] = tile + ((rvx && ((pvalid >> h) & 1)) ? oX + h * pitch : p.zslot);
            if constexpr (BACKWARD) {
                uint32_t xnw[NA][5];
                read_rows<T, NA>(xs, rx, m2, ji, xnw);
                if (t >= 1) {
                    // weight-gradient sums: g * corner differences (corner_diffs, shiftnd_common.hpp) with the
                    // differences that neighbouring elements / rows share computed once
                    const u4 gq = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(tile + oG, 16));
                    Chunk<S, E> gch;
                    __builtin_memcpy(gch.e, &gq, 16);
                    if constexpr (ES == 2) {
                        uint32_t gp[4], gs[5];  // g pairs [g(2i), g(2i+1)] and the same shifted by one: [g(2i-1), g(2i)]
                        __builtin_memcpy(gp, gch.e, 16);
                        gs[0] = gp[0] << 16;
#pragma unroll
                        for (int i = 1; i < 4; ++i) gs[i] = __builtin_amdgcn_alignbit(gp[i], gp[i - 1], 16);
                        gs[4] = gp[3] >> 16;
#pragma unroll
                        for (int r = 0; r < 2 * NA; ++r) {
                            const uint32_t(&wdw)[5] = r < NA ? xpw[r < NA ? r : 0] : xnw[r < NA ? 0 : r - NA];
#pragma unroll
                            for (int i = 0; i < 4; ++i) part[2 * r] = dot2_acc<T>(gp[i], wdw[i], part[2 * r]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) part[2 * r + 1] = dot2_acc<T>(gs[i], wdw[i], part[2 * r + 1]);
                            part[2 * r + 1] = dot2_acc<T>(gs[4], wdw[4] & 0xffffu, part[2 * r + 1]);
                        }
                    } else {
                    CT xp[NA][E + 1], xn[NA][E + 1];
#pragma unroll
                    for (int h = 0; h < NA; ++h) {
                        unpack_window<T>(xpw[h], xp[h]);
                        unpack_window<T>(xnw[h], xn[h]);
                    }
                    if constexpr (ND == 3) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const CT gval = widen<T>(gch.e[e]);
                            part[0] = fma_ct(gval, xn[0][e] - xp[0][e], part[0]);
                            part[1] = fma_ct(gval, xn[1][e] - xp[1][e], part[1]);
                            part[2] = fma_ct(gval, xn[0][e + 1] - xp[0][e + 1], part[2]);
                            part[3] = fma_ct(gval, xn[1][e + 1] - xp[1][e + 1], part[3]);
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
                    }
                    if ((t & (kFlushSteps - 1)) == 0) flush();
                }
#pragma unroll
                for (int h = 0; h < NA; ++h)
#pragma unroll
                    for (int i = 0; i < 5; ++i) xpw[h][i] = xnw[h][i];
                // the two halves of a step are independent: keep the scheduler from overlapping them (it would hold
                // both halves' rows in registers at once and cost a wave of occupancy)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (ACTIVE) {
                uint32_t gw[NA][5];
                if constexpr (BACKWARD) {
                    const bool rvg = g1[(ND == 3 ? ubstart : mybstart) + t] >= 0;
                    const char *gsl[NA];
#pragma unroll
                    for (int h = 0; h < NA; ++h) gsl[h] = tile + ((rvg && ((pvalid >> (2 + h)) & 1)) ? oGS + h * pitch : p.zslot);
                    read_rows<T, NA>(gsl, rg, g2, ji, gw);
                } else {
                    read_rows<T, NA>(xs, rx, m2, ji, gw);
                }
                CT ln[E + 1];
                {
                    CT gn[NA][E + 1];
#pragma unroll
                    for (int h = 0; h < NA; ++h) unpack_window<T>(gw[h], gn[h]);
#pragma unroll
                    for (int e = 0; e <= E; ++e) {
                        if constexpr (ND == 3) ln[e] = lerp_t<T>(gn[0][e], gn[1][e], dA);
                        else ln[e] = gn[0][e];
                    }
                }
                if (t >= 1) {
                    CT m[E + 1];
#pragma unroll
                    for (int e = 0; e <= E; ++e) m[e] = lerp_t<T>(lp[e], ln[e], dB);
                    Chunk<S, E> res;
#pragma unroll
                    for (int e = 0; e < E; ++e) res.e[e] = narrow<T>(lerp_t<T>(m[e], m[e + 1], dI));
                    pres = res;
                    prow = mybstart + t - 1;
                }
#pragma unroll
                for (int e = 0; e <= E; ++e) lp[e] = ln[e];
            } else {
                if (t >= 1) {  // the sparse shift's input gradient is a copy of the shifted grad_out row
                    uint32_t gw[1][5];
                    const bool rvg = g1[(ND == 3 ? ubstart : mybstart) + t - 1] >= 0;
                    const char *gsl[1] = {tile + ((rvg && ((pvalid >> 2) & 1)) ? oGS : p.zslot)};
                    read_rows<T, 1>(gsl, rg, g2, ji, gw);
                    Chunk<S, E> res;
                    __builtin_memcpy(res.e, gw[0], 16);
                    pres = res;
                    prow = mybstart + t - 1;
                }
            }
        }
};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_loads(d, pv[d]);
    int t0 = 0;
    for (; t0 + DEPTH <= p.seg + 1; t0 += DEPTH) {   // whole groups: nothing conditional between the steps (exact wait counts)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) step(t0 + d, pv[d]);
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (t0 + d <= p.seg) step(t0 + d, pv[d]);
    flush_store();

    if constexpr (BACKWARD) {
        flush();
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
    auto lds_of = [&](int n) { return 2 * (static_cast<size_t>(slots_of(n) + 2) * (RB + kGuard) + kGuard) + map_bytes; };
    while (nseg > 1 && (static_cast<int64_t>(slots_of(nseg)) * pl.cpr > static_cast<int64_t>(np_max) * kThreads ||
                        lds_of(nseg) > 64 * 1024))
        --nseg;
    if (static_cast<int64_t>(slots_of(nseg)) * pl.cpr > static_cast<int64_t>(np_max) * kThreads) return pl;
    (void)kinds;
    // workgroups wanted: whole rounds of what the chip holds at once (256 CUs x 4 resident workgroups) and few, long
    // bands (a band pays one warm-up step).  Measured on C3 (tools/kbench.py --knobs 13=...): backward 3072 -> 0.306 ms
    // (2048: 0.314, 4096: 0.317, 8192: 0.326), forward 1024 -> 0.187 ms (2048: 0.205, 4096: 0.209)
    const int64_t min_wgs = g_slide_tune[1] > 0 ? g_slide_tune[1] : (backward ? 3072 : 1024);
    const int64_t min_rows = g_slide_tune[2] > 0 ? g_slide_tune[2] : 16;
    if (g.nd == 3) {
        if (plane * es >= (1LL << 31)) return pl;  // buffer offsets of the output store
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
        if (span >= (1LL << 30) || span * es >= (1LL << 31)) return pl;
    }
    pl.nseg = nseg;
    pl.nslots = slots_of(nseg);
    pl.npieces = pl.nslots * pl.cpr;
    pl.pitch = static_cast<int>(RB) + kGuard;  // [guard][row]: the next slot's guard (or the tile's tail) is the right guard
    pl.tile_bytes = (pl.nslots + 2) * pl.pitch + kGuard;  // + the all-zero slot, the dump slot and the tail guard
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
    p.zslot = pl.nslots * pl.pitch;
    p.npieces = pl.npieces;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(pl.cpr));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(pl.bands));
    p.d_inner = make_fastdiv(static_cast<uint32_t>(pl.inner));
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    for (int d = 0; d < 3; ++d) p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(static_cast<int>(g.S[d]), g.pad)));
}

constexpr int kNpBackward = 3, kNpForward = 2;
// steps of staging in flight: 2 costs 12 (backward) registers and a wave of occupancy, measured slower (C3 backward
// 0.337 -> 0.356 ms)
// staged rows in flight.  With every memory instruction of the walk unconditional the waits are exact, and a second
// step of loads in flight pays for 16-bit data (C3 bf16 backward 0.301 -> 0.292 ms, the 3-D sparse shift 0.306 -> 0.293);
// fp32 (twice the registers per piece) and the forward measure flat or worse, three steps worse everywhere.
template <typename T> constexpr int kDepthBackward = sizeof(typename T::S) == 2 ? 2 : 1;
constexpr int kDepthForward = 1;

template <typename T, bool ACTIVE>
void launch_slide_backward(const SlideParams &p, const SlidePlan &pl, hipStream_t st) {
    if (p.nd == 3) hipLaunchKernelGGL((slide_kernel<T, 3, ACTIVE, true, kNpBackward, kDepthBackward<T>>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    else hipLaunchKernelGGL((slide_kernel<T, 2, ACTIVE, true, kNpBackward, kDepthBackward<T>>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
}
template <typename T>
void launch_slide_forward(const SlideParams &p, const SlidePlan &pl, hipStream_t st) {
    if (p.nd == 3) hipLaunchKernelGGL((slide_kernel<T, 3, true, false, kNpForward, kDepthForward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    else hipLaunchKernelGGL((slide_kernel<T, 2, true, false, kNpForward, kDepthForward>), dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
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
    const int groups = pl.groups * pl.inner;
#define SHIFTND_SLIDE_BWD(TT) \
    { \
        if (g.active) launch_slide_backward<TT, true>(p, pl, st); \
        else launch_slide_backward<TT, false>(p, pl, st); \
        reduce_weight_grads_of<TT>(p.partials, groups, p.C, p.nd, gw, st); \
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
