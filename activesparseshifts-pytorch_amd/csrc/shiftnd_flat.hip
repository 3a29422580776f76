// shiftnd_flat.hip -- contiguous 1-D / 2-D problems whose rows are NOT whole 16-byte pieces, as one-step workgroups over the
// tensor's FLAT stream of 16-byte chunks; gfx950 (MI355X), round 5.  DESIGN section 3.19.
//
// The shapes: the deep stages of every ImageNet network (14 x 14 and 7 x 7 planes: 56- / 28-byte fp32 rows), odd image sizes
// (113, 225, 299 ...), and the OUTPUT of every cropped shift as the next layer's input (62 x 62, 222 x 222: the reference's
// emulate_dw modules cut one element per side, modules/shifts.py:41-46, tests/shifts_test.py:9-28).  The chunk kernels need rows of
// whole pieces; until round 5 these shapes ran per-channel kernels with element-wide loads at 2 - 3 TB/s (shiftnd_small.hip).
//
// A contiguous NC[H]W tensor is one stream of planes: plane p = (n, c) starts at byte p x plane_bytes, at no particular alignment,
// but the TENSOR starts 16-byte aligned.  A workgroup owns ONE step = 256 consecutive chunks (4 KB) of the flat output (forward) /
// grad_x (backward) stream and the grid is every step in memory order (XCD k owns the k-th eighth: shiftnd_step.hip's rule).
//   * small planes (the source planes of a step fit the LDS budget): the step's planes p_a .. p_b of the SOURCE tensor(s) are one
//     contiguous byte range -- staged by LDS-DMA (global_load_lds, 16-byte pieces, lane-linear: no decode at all), whatever the
//     channels' shifts are; every padding's sources lie inside (a plane is its own padding domain);
//   * large planes (a step touches at most two): per plane the contiguous range of source ROWS the step's rows read under zeros /
//     border padding (the unfolded row range clamped into the plane); rows the periodic / reflecting paddings fold elsewhere --
//     a few rows at the top and bottom of a plane -- are read from memory by the elements that need them;
//   * a thread produces one output chunk = E elements that may straddle rows and planes: (plane, row, column) of the first
//     element by two multiply-shift divisions, the others by carries; every element is a map lookup (arithmetic: fold_index) and
//     an LDS read at its own address -- the same code for every position, no divergent edge path;
//   * per-plane parameters (canonical shifts, fractions, LDS bases) sit in a small LDS table filled by the first threads while
//     the DMA is in flight; a chunk reads its plane's entry and the next one's.
// Backward: reference semantics of kernels/shifts_kernels.h:222-327 (the window: :271, :295-297, :314-324).  Weight gradients in
// the multilinear form (shiftnd_common.hpp): per chunk and plane two fp32 sums of g x corner difference, parked in LDS, added per
// plane by 16-lane groups in a fixed order (fp64) into one record per (step, plane); flat_reduce adds a channel's records in a
// fixed order and applies the blends once.  Deterministic, no atomics.
// Forward: kernels/shifts_kernels.h:156-220.  Roofline: HBM, 2 x s bytes per element forward, 3 x s backward.
#include "shiftnd_step.hpp"

namespace shiftnd {
namespace {

constexpr int kMaxPlanes = 72;             // planes a step may touch (small planes: 4096 / plane bytes + 2)
constexpr int kCoverBudget = 24 * 1024;    // LDS bytes per staged tensor

struct FlatDesc {   // per channel, written by flat_prep (backward)
    int cx1, cx2;   // canonical shifts of the input's row / column maps (sizes S1, S2)
    int cg1, cg2;   // ... of the gradient's maps (the window's sizes O1, O2; sparse shift: the opposite direction)
    double dw[2];   // fractions of prep_shift_backward (row, column), exactly as the compute type holds them
};

struct FlatParams {
    const void *x;        // forward: input; backward: saved input
    const void *go;       // backward: incoming gradient
    void *out;            // forward: output; backward: grad_x
    const void *w;
    FlatDesc *desc;       // backward: [C]
    double *partials;     // backward: [total_steps + planes][2]
    int wkind, C, pad, nd;
    int S1, S2, O1, O2, L1, L2;
    uint32_t XP, OP;      // elements per plane of x / of out (grad_out)
    uint32_t total;       // elements of the streamed tensor (forward: out; backward: grad_x)
    uint32_t planes;      // N * C
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_SP, d_SR;   // the streamed tensor's plane elements / row length
    FastDiv d_C, d_per1, d_per2, d_pero1, d_pero2;
    int lds_bytes;        // the launch's dynamic LDS size (clamps the corner reads of the zeros-padding forms)
    int front;            // bytes of slack between the tables and the first cover (a corner group that starts one row / column early)
};

template <typename S> __device__ __forceinline__ S lds_at(const char *smem, int off) { return *reinterpret_cast<const S *>(smem + off); }

constexpr int kZero = 0;   // LDS bytes [0, 16) hold zeros: what an element outside the window / in the padding reads (an address select, no
                           // select on the value and no exec-mask region around the read: a kernel this short lives or dies by its control flow)
constexpr int kTab = 16;   // the plane table behind them

// canon_shift (shiftnd_common.hpp) of an integral shift held in the compute type: 32-bit arithmetic below 2^30 (always, in practice),
// the 64-bit form beyond; the padding mode is a run-time value
template <typename CT> __device__ __forceinline__ int canon_rt(CT r, int len, int pad, const FastDiv &dper) {
    if (len <= 1) return 0;
    if (r > CT(-1073741824) && r < CT(1073741824)) {
        const int s = static_cast<int>(r);
        if (pad <= 1) return s < -len - 1 ? -len - 1 : (s > len + 1 ? len + 1 : s);
        const int period = pad == 2 ? len : (pad == 3 ? 2 * (len - 1) : 2 * len);
        const uint32_t a = static_cast<uint32_t>(s < 0 ? -s : s);
        const uint32_t m = a - fdiv(a, dper) * static_cast<uint32_t>(period);
        return static_cast<int>((s < 0 && m != 0) ? static_cast<uint32_t>(period) - m : m);
    }
    return canon_shift(static_cast<int64_t>(r), len, pad, dper);
}

// fold_index for PAD = kPadRT (shiftnd_step.hpp): the wrapping paddings 2 .. 4 through ONE instantiation, the mode a kernel argument;
// both folds are (idx ^ m) + c with launch-uniform m (0 periodic, -1 reflect / symmetric) and c (shiftnd_step.hpp fold_index_rt
// without the border mode, which keeps its own instantiation here: it clamps into the staged rows)
constexpr int kPadRT = 5;   // template value of the flat-stream kernels: periodic, reflect or symmetric, the mode in p.pad
struct FoldRT { int m, cn, ch; };
__device__ __forceinline__ FoldRT fold_coeffs(int len, int pad) {
    FoldRT f;
    f.m = pad >= 3 ? -1 : 0;
    f.cn = pad == 2 ? len : (pad == 3 ? 1 : 0);
    f.ch = pad == 2 ? -len : (pad == 3 ? 2 * (len - 1) + 1 : 2 * len);
    return f;
}
template <int PAD> __device__ __forceinline__ int fold_t(int idx, int len, const FoldRT &f) {
    if constexpr (PAD == kPadRT) {
        const int t = idx < 0 ? (idx ^ f.m) + f.cn : idx;
        return t > len - 1 ? (t ^ f.m) + f.ch : t;
    } else {
        return fold_index(idx, len, PAD);
    }
}

__device__ __forceinline__ int imap(int p, int cs, int len, int pad) { return len == 1 ? 0 : fold_index(p - cs, len, pad); }

// one plane's parameters as the element loop reads them
template <typename CT> struct PlaneRec {
    int c1, c2;      // canonical shifts of the map the kernel gathers the FIRST staged tensor through (x)
    int g1, g2;      // backward: ... the gradient through
    int xb, gb;      // LDS byte offset of the plane's element (0, 0) in the staged cover of x / of the gradient (may be negative: rows before the cover)
    int xr0, xr1;    // staged rows of x (large planes; small planes: 0 .. S1 - 1)
    int gr0, gr1;    // ... of the gradient
    CT f1, f2;       // fractions (row, column)
};

// cover of rows [r0, r1] of plane pl (elements per plane P, row length R): 16-byte pieces [first, first + n) of the tensor
struct Cover {
    uint32_t first;   // first piece (byte offset >> 4)
    int n;            // pieces (0: nothing)
    int base;         // LDS byte offset of the plane's element (0, 0), relative to the cover's first staged byte
};
template <int ES> __device__ __forceinline__ Cover row_cover(uint32_t pl, uint32_t P, int R, int r0, int r1) {
    Cover c;
    if (r1 < r0) {
        c.first = 0;
        c.n = 0;
        c.base = 0;
        return c;
    }
    const uint64_t b0 = (static_cast<uint64_t>(pl) * P + static_cast<uint64_t>(r0) * R) * ES, b1 = (static_cast<uint64_t>(pl) * P + static_cast<uint64_t>(r1 + 1) * R) * ES;
    c.first = static_cast<uint32_t>(b0 >> 4);
    c.n = static_cast<int>(((b1 + 15) >> 4) - (b0 >> 4));
    c.base = static_cast<int>(b0 & 15) - r0 * R * ES;
    return c;
}

// LDS-DMA of pieces [first, first + n) of `src` to smem + dst (16-byte aligned), lane-linear; every thread of the workgroup calls it.
// The last piece of a tensor whose size is not a multiple of 16 reaches up to 15 bytes past its end: inside the same aligned
// 16-byte granule as its last valid byte, hence inside the same page -- never a fault; the bytes are never used.
__device__ __forceinline__ void stage(const void *src, uint32_t first, int n, char *smem, int dst) {
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char *base = static_cast<const char *>(src) + (static_cast<uint64_t>(first) << 4);
    for (int k = 0; k * kThreads < n; ++k) {   // (uniform trip count)
        const int q = k * kThreads + tid;
        if (q < n) {
            char *dst_wave = smem + dst + (k * kThreads + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + static_cast<uint32_t>(q) * 16u),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    }
}

// ... by the first wave alone (large planes: the covers depend on the channel's shift; the scalar work that leads to them -- and the CU
// has ONE scalar unit -- is done by one wave of the four instead of all of them)
__device__ __forceinline__ void stage_wave0(const void *src, uint32_t first, int n, char *smem, int dst) {
    const int lane = static_cast<int>(threadIdx.x);   // (called under threadIdx.x < 64)
    const char *base = static_cast<const char *>(src) + (static_cast<uint64_t>(first) << 4);
    for (int k = 0; k * 64 < n; ++k) {
        const int q = k * 64 + lane;
        if (q < n) {
            char *dst_wave = smem + dst + k * 64 * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + static_cast<uint32_t>(q) * 16u),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    }
}

// the unfolded source rows [a, b] of a plane clamped into it (exact for zeros and border padding); empty when zeros padding puts
// every row outside
__device__ __forceinline__ void clamp_rows(int a, int b, int len, bool zeros, int &r0, int &r1) {
    if (zeros && (b < 0 || a >= len)) {
        r0 = 0;
        r1 = -1;
        return;
    }
    r0 = a < 0 ? 0 : (a >= len ? len - 1 : a);
    r1 = b < 0 ? 0 : (b >= len ? len - 1 : b);
}

// ---------------------------------------------------------------------------------------------------------------------
// flat_forward<T, ACTIVE, PAD, SMALL>: PAD = 0 zeros, 1 border, kPadRT = the three wrapping modes with the mode a kernel argument and
// a branch-free fold (every element folds its coordinates: through a run-time `switch` the reflecting paddings ran at half the zeros
// padding's rate); SMALL = the whole source planes of a step are staged.
// LDS: [plane table][cover of x]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool ACTIVE, int PAD, bool SMALL>
__global__ __launch_bounds__(kThreads) void flat_forward(const FlatParams p) {
    constexpr bool PADZ = PAD == 0;
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    using Rec = PlaneRec<CT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rec *table = reinterpret_cast<Rec *>(smem + kTab);
    constexpr int TAB = kTab + (((SMALL ? kMaxPlanes : 2) * static_cast<int>(sizeof(Rec)) + 15) & ~15);
    const int COV = TAB + p.front;   // LDS offset of the cover
    if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(smem)[threadIdx.x] = 0u;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const int tid = static_cast<int>(threadIdx.x);
    const int S1 = p.S1, S2 = p.S2, O1 = p.O1, O2 = p.O2, L1 = p.L1, L2 = p.L2;
    const int pad = PAD == kPadRT ? p.pad : PAD;
    const FoldRT fS1 = fold_coeffs(S1, pad), fS2 = fold_coeffs(S2, pad);
    (void)fS1;
    (void)fS2;
    const uint32_t f0 = bid * static_cast<uint32_t>(kThreads * E);                                  // first element of the step
    const uint32_t f1 = min(p.total, f0 + static_cast<uint32_t>(kThreads * E)) - 1u;                // ... and its last
    const uint32_t plA = fdiv(f0, p.d_SP), plB = fdiv(f1, p.d_SP);
    const int nplanes = static_cast<int>(plB - plA) + 1;

    auto channel = [&](uint32_t pl, int &c1, int &c2, CT &fr1, CT &fr2) {   // the plane's shifts (shifts_cpu.cpp:223-224), canonical
        const int c = static_cast<int>(pl - fdiv(pl, p.d_C) * static_cast<uint32_t>(p.C));
        // (weights of the tensor's dtype -- shiftnd_api.hip routes nothing else here -- but T only says the element SIZE for the sparse shift)
        const CT w1 = p.nd == 2 ? load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 2) : CT(0);
        const CT w2 = load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd + p.nd - 1);
        const CT r1 = ACTIVE ? c_floor<CT>(w1) : c_rint<CT>(w1), r2 = ACTIVE ? c_floor<CT>(w2) : c_rint<CT>(w2);
        fr1 = ACTIVE ? w1 - r1 : CT(0);
        fr2 = ACTIVE ? w2 - r2 : CT(0);
        c1 = canon_rt<CT>(r1, S1, pad, p.d_per1);
        c2 = canon_rt<CT>(r2, S2, pad, p.d_per2);
    };

    if constexpr (SMALL) {
        const Cover cv = row_cover<ES>(plA, p.XP, S2, 0, static_cast<int>((plB - plA + 1) * static_cast<uint32_t>(S1)) - 1);   // planes plA .. plB whole
        stage(p.x, cv.first, cv.n, smem, COV);
        if (tid < nplanes) {
            Rec r;
            channel(plA + tid, r.c1, r.c2, r.f1, r.f2);
            r.xb = COV + cv.base + tid * static_cast<int>(p.XP) * ES;
            r.xr0 = 0;
            r.xr1 = S1 - 1;
            r.g1 = r.g2 = r.gb = r.gr0 = r.gr1 = 0;
            table[tid] = r;
        }
    } else {
        // at most two planes; the first wave evaluates them, stages their covers and writes the table
        int used = 0;
        if (tid < 64)
        for (int k = 0; k < nplanes; ++k) {   // (uniform trip count: one plane for all but the steps at a plane's end)
            const uint32_t pl = plA + k;
            Rec r;
            channel(pl, r.c1, r.c2, r.f1, r.f2);
            // output rows of this plane inside the step
            const uint32_t lo = k == 0 ? f0 - plA * p.OP : 0u, hi = (k == 0 && plB == plA) || k == 1 ? f1 - pl * p.OP : p.OP - 1u;
            const int i0 = static_cast<int>(fdiv(lo, p.d_SR)), i1 = static_cast<int>(fdiv(hi, p.d_SR));
            int r0 = 0, r1 = -1;
            clamp_rows(S1 == 1 ? 0 : i0 + L1 - r.c1, S1 == 1 ? 0 : i1 + L1 - r.c1 + (ACTIVE ? 1 : 0), S1, PADZ, r0, r1);
            const Cover cv = row_cover<ES>(pl, p.XP, S2, r0, r1);
            stage_wave0(p.x, cv.first, cv.n, smem, COV + used);
            r.xb = COV + used + cv.base;
            r.xr0 = r0;
            r.xr1 = r1;
            r.g1 = r.g2 = r.gb = r.gr0 = r.gr1 = 0;
            used += cv.n * 16;
            if (tid == k) table[k] = r;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t f = f0 + static_cast<uint32_t>(tid * E);
    if (f >= p.total) return;
    const uint32_t pl = fdiv(f, p.d_SP);
    const uint32_t r = f - pl * p.OP;
    int i = static_cast<int>(fdiv(r, p.d_SR));
    int j = static_cast<int>(r) - i * O2;
    const int slot = static_cast<int>(pl - plA);
    const Rec D0 = table[slot], D1 = table[min(slot + 1, nplanes - 1)];
    const int ecut = static_cast<int>(min(static_cast<uint32_t>(E), p.OP - r));   // elements of this chunk in plane pl
    const S *xg = static_cast<const S *>(p.x);
    const S zero = static_cast<S>(0.0f);
    // the last LDS byte offset a corner group may start at (the launch adds a row of slack behind the covers: every VALID corner
    // group starts below it, only groups of masked corners are moved)
    const int lim = p.lds_bytes - ((S1 == 1 ? 0 : S2) + 2) * ES;
    const bool one_d = p.nd == 1;                  // (uniform) Shift1d: interp1D of the two column corners (interpolation.h:3-7)
    Chunk<S, E> res;
    // SIMPLE: no chunk of this wave changes planes and every source row it folds to is staged -- no per-element selects between two
    // plane records, no staged-row checks (large planes: all but the waves at a plane's first / last rows)
    auto elements = [&](auto simple_tag) {
    constexpr bool SIMPLE = decltype(simple_tag)::value;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool first = SIMPLE || e < ecut;
        const int c1 = first ? D0.c1 : D1.c1, c2 = first ? D0.c2 : D1.c2, xb = first ? D0.xb : D1.xb;
        const int a = S1 == 1 ? 0 : i + L1 - c1, b = S2 == 1 ? 0 : j + L2 - c2;   // the source element, unfolded coordinates
        if constexpr (PADZ) {
            const bool ra = static_cast<unsigned>(a) < static_cast<unsigned>(S1), cb = static_cast<unsigned>(b) < static_cast<unsigned>(S2);
            if constexpr (!ACTIVE) {
                res.e[e] = lds_at<S>(smem, (ra && cb) ? xb + (a * S2 + b) * ES : kZero);   // (an address select: the zero words; no exec region)
            } else {
                // the four corners from two addresses (row a, row a + 1; columns b, b + 1 adjacent), read unconditionally at a clamped
                // address and masked afterwards
                const bool ra1 = S1 == 1 ? ra : static_cast<unsigned>(a + 1) < static_cast<unsigned>(S1);
                const bool cb1 = S2 == 1 ? cb : static_cast<unsigned>(b + 1) < static_cast<unsigned>(S2);
                const int base = min(max(xb + (a * S2 + b) * ES, 0), lim);
                const int dn = S1 == 1 ? 0 : S2 * ES, rt = S2 == 1 ? 0 : ES;
                const S q00 = lds_at<S>(smem, base), q01 = lds_at<S>(smem, base + rt), q10 = lds_at<S>(smem, base + dn), q11 = lds_at<S>(smem, base + dn + rt);
                const CT fr[2] = {first ? D0.f1 : D1.f1, first ? D0.f2 : D1.f2};
                const CT v00 = (ra && cb) ? widen<T>(q00) : CT(0), v10 = (ra1 && cb) ? widen<T>(q10) : CT(0);
                const CT v01 = (ra && cb1) ? widen<T>(q01) : CT(0), v11 = (ra1 && cb1) ? widen<T>(q11) : CT(0);
                if (one_d) {
                    const CT v[2] = {v00, v01};
                    res.e[e] = narrow<T>(interp_t<T, 1>(v, fr + 1));
                } else {
                    const CT v[4] = {v00, v10, v01, v11};   // corner order: bit 0 = + 1 row, bit 1 = + 1 column (shifts_kernels.h:58-103)
                    res.e[e] = narrow<T>(interp_t<T, 2>(v, fr));
                }
            }
        } else {
            const uint32_t ple = first ? pl : pl + 1u;
            const int xr0 = first ? D0.xr0 : D1.xr0, xr1 = first ? D0.xr1 : D1.xr1;
            auto tap = [&](int ar, int bc) -> S {   // folded coordinates: always a source element
                if constexpr (SMALL || PAD == 1 || SIMPLE) {   // (border padding clamps into the staged rows)
                    return lds_at<S>(smem, xb + (ar * S2 + bc) * ES);
                } else {
                    if (ar >= xr0 && ar <= xr1) return lds_at<S>(smem, xb + (ar * S2 + bc) * ES);
                    return xg[static_cast<uint64_t>(ple) * p.XP + static_cast<uint32_t>(ar * S2 + bc)];   // a row the padding folds out of the cover
                }
            };
            // (SIMPLE on large planes: the chunk's rows lie inside the plane unfolded)
            constexpr bool ROWS_IN = SIMPLE && !SMALL && PAD >= 2;
            const int ar = S1 == 1 ? 0 : (ROWS_IN ? a : fold_t<PAD>(a, S1, fS1)), bc = S2 == 1 ? 0 : fold_t<PAD>(b, S2, fS2);
            if constexpr (!ACTIVE) {
                res.e[e] = tap(ar, bc);
            } else {
                const int ar1 = S1 == 1 ? 0 : (ROWS_IN ? a + 1 : fold_t<PAD>(a + 1, S1, fS1)), bc1 = S2 == 1 ? 0 : fold_t<PAD>(b + 1, S2, fS2);
                const CT fr[2] = {first ? D0.f1 : D1.f1, first ? D0.f2 : D1.f2};
                if (one_d) {
                    const CT v[2] = {widen<T>(tap(ar, bc)), widen<T>(tap(ar, bc1))};
                    res.e[e] = narrow<T>(interp_t<T, 1>(v, fr + 1));
                } else {
                    const CT v[4] = {widen<T>(tap(ar, bc)), widen<T>(tap(ar1, bc)), widen<T>(tap(ar, bc1)), widen<T>(tap(ar1, bc1))};
                    res.e[e] = narrow<T>(interp_t<T, 2>(v, fr));
                }
            }
        }
        ++j;
        if (j == O2) {
            j = 0;
            ++i;
            if (i == O1) i = 0;
        }
    }
    };
    // rows this chunk reads, unfolded: [i + L1 - c1, i_last + L1 - c1 (+ 1)] -- inside the plane: nothing folds, everything is staged
    bool simple = ecut == E;
    if constexpr (!SMALL) {
        const int ilast = static_cast<int>(fdiv(static_cast<uint32_t>(r) + E - 1, p.d_SR));
        const int a_lo = i + L1 - D0.c1, a_hi = ilast + L1 - D0.c1 + (ACTIVE ? 1 : 0);
        if constexpr (PAD >= 2) simple = simple && (S1 == 1 || (a_lo >= 0 && a_hi < S1));
    }
    if (__all(simple)) elements(std::true_type{});
    else elements(std::false_type{});
    (void)zero;
    S *op = static_cast<S *>(p.out) + f;
    if (f + E <= p.total) {
        store_chunk<S, E>(op, res);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (f + e < p.total) op[e] = res.e[e];
    }
}

// The weight preparation of the reference's backward (shifts_cpu.cpp:242-244) for channel c, in the kernel that needs it (round 5: a
// prep kernel of its own cost a 60 us call 5 us): fractions, canonical shifts of the input's maps (sizes S) and of the gradient's
// (the window's sizes O; sparse shift: the opposite direction).
template <typename CT, bool ACTIVE>
__device__ __forceinline__ void flat_channel_backward(const FlatParams &p, int c, int &cx1, int &cx2, int &cg1, int &cg2, CT &f1, CT &f2) {
    const CT w1 = p.nd == 2 ? load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 2) : CT(0);
    const CT w2 = load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd + p.nd - 1);
    auto frac = [](CT w) { return ACTIVE ? (w - c_floor<CT>(w)) : ((w > CT(0)) ? (w - c_floor<CT>(w)) : (c_ceil<CT>(w) - w)); };   // prep_shift_backward
    f1 = p.nd == 2 ? frac(w1) : CT(0);
    f2 = frac(w2);
    const CT r1 = ACTIVE ? (w1 - f1) : c_rint<CT>(w1), r2 = ACTIVE ? (w2 - f2) : c_rint<CT>(w2);   // integral
    cx1 = canon_rt<CT>(r1, p.S1, p.pad, p.d_per1);
    cx2 = canon_rt<CT>(r2, p.S2, p.pad, p.d_per2);
    cg1 = canon_rt<CT>(ACTIVE ? r1 : -r1, p.O1, p.pad, p.d_pero1);
    cg2 = canon_rt<CT>(ACTIVE ? r2 : -r2, p.O2, p.pad, p.d_pero2);
}

// flat_prep: one thread per channel writes the channel's descriptor.  (Computing it in flat_backward itself -- every workgroup for
// each plane it touches, tried in round 5 to save the launch -- puts weight loads and the shift arithmetic in front of a one-step
// workgroup's first DMA: FLAT_INLINE_PREP=1 builds that form for A / B runs.)
#ifndef FLAT_INLINE_PREP
#define FLAT_INLINE_PREP 0
#endif
template <typename T>
__global__ __launch_bounds__(kThreads) void flat_prep(const FlatParams p, const int active) {
    using CT = typename T::C;
    const int c = static_cast<int>(blockIdx.x * kThreads + threadIdx.x);
    if (c >= p.C) return;
    FlatDesc d;
    CT f1, f2;
    if (active) flat_channel_backward<CT, true>(p, c, d.cx1, d.cx2, d.cg1, d.cg2, f1, f2);
    else flat_channel_backward<CT, false>(p, c, d.cx1, d.cx2, d.cg1, d.cg2, f1, f2);
    d.dw[0] = static_cast<double>(f1);
    d.dw[1] = static_cast<double>(f2);
    p.desc[c] = d;
}

// ---------------------------------------------------------------------------------------------------------------------
// flat_backward<T, ACTIVE, PADZ, SMALL>.  LDS: [plane table][per-chunk partial sums][cover of x][cover of grad_out]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool ACTIVE, int PAD, bool SMALL>
__global__ __launch_bounds__(kThreads) void flat_backward(const FlatParams p, int gcov_off) {
    constexpr bool PADZ = PAD == 0;
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    using Rec = PlaneRec<CT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rec *table = reinterpret_cast<Rec *>(smem + kTab);
    constexpr int TAB = kTab + (((SMALL ? kMaxPlanes : 2) * static_cast<int>(sizeof(Rec)) + 15) & ~15);
    if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(smem)[threadIdx.x] = 0u;
    constexpr int PART = TAB;                                        // [256][4] CT: (sumA, sumB) of the chunk's first plane, of its second
    const int XCOV = PART + kThreads * 4 * static_cast<int>(sizeof(CT)) + p.front;
    const int GCOV = XCOV + gcov_off;                                // (host: the largest x cover of any step, 16-byte aligned)

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const int tid = static_cast<int>(threadIdx.x);
    const int S1 = p.S1, S2 = p.S2, O1 = p.O1, O2 = p.O2, L1 = p.L1, L2 = p.L2;
    const FoldRT fS1 = fold_coeffs(S1, p.pad), fS2 = fold_coeffs(S2, p.pad), fO1 = fold_coeffs(O1, p.pad), fO2 = fold_coeffs(O2, p.pad);
    (void)fS1; (void)fS2; (void)fO1; (void)fO2;
    const uint32_t f0 = bid * static_cast<uint32_t>(kThreads * E);
    const uint32_t f1 = min(p.total, f0 + static_cast<uint32_t>(kThreads * E)) - 1u;
    const uint32_t plA = fdiv(f0, p.d_SP), plB = fdiv(f1, p.d_SP);
    const int nplanes = static_cast<int>(plB - plA) + 1;
    const bool cropped = O1 != S1 || O2 != S2;   // (uniform)

    auto fill = [&](uint32_t pl, Rec &r) {
        const int c = static_cast<int>(pl - fdiv(pl, p.d_C) * static_cast<uint32_t>(p.C));
#if FLAT_INLINE_PREP
        flat_channel_backward<CT, ACTIVE>(p, c, r.c1, r.c2, r.g1, r.g2, r.f1, r.f2);
#else
        const FlatDesc d = p.desc[c];
        r.c1 = d.cx1;
        r.c2 = d.cx2;
        r.g1 = d.cg1;
        r.g2 = d.cg2;
        r.f1 = static_cast<CT>(d.dw[0]);
        r.f2 = static_cast<CT>(d.dw[1]);
#endif
    };
    if constexpr (SMALL) {
        const Cover cx = row_cover<ES>(plA, p.XP, S2, 0, static_cast<int>((plB - plA + 1) * static_cast<uint32_t>(S1)) - 1);
        const Cover cg = row_cover<ES>(plA, p.OP, O2, 0, static_cast<int>((plB - plA + 1) * static_cast<uint32_t>(O1)) - 1);
        stage(p.x, cx.first, cx.n, smem, XCOV);
        stage(p.go, cg.first, cg.n, smem, GCOV);
        if (tid < nplanes) {
            Rec r;
            fill(plA + tid, r);
            r.xb = XCOV + cx.base + tid * static_cast<int>(p.XP) * ES;
            r.gb = GCOV + cg.base + tid * static_cast<int>(p.OP) * ES;
            r.xr0 = 0;
            r.xr1 = S1 - 1;
            r.gr0 = 0;
            r.gr1 = O1 - 1;
            table[tid] = r;
        }
    } else {
        int usedx = 0, usedg = 0;
        if (tid < 64)   // (the first wave alone: see stage_wave0)
        for (int k = 0; k < nplanes; ++k) {   // (uniform trip count: one plane for all but the steps at a plane's end)
            const uint32_t pl = plA + k;
            Rec r;
            fill(pl, r);
            const uint32_t lo = k == 0 ? f0 - plA * p.XP : 0u, hi = (k == 0 && plB == plA) || k == 1 ? f1 - pl * p.XP : p.XP - 1u;
            const int i0 = static_cast<int>(fdiv(lo, p.d_SR)), i1 = static_cast<int>(fdiv(hi, p.d_SR));
            int xr0 = 0, xr1 = -1, gr0 = 0, gr1 = -1;
            {
                // rows of the window inside the step's rows (elements outside the window read nothing)
                const int w0 = max(i0, L1), w1 = min(i1, L1 + O1 - 1);
                if (w0 <= w1) {
                    clamp_rows(S1 == 1 ? 0 : w0 - r.c1, S1 == 1 ? 0 : w1 - r.c1 + 1, S1, PADZ, xr0, xr1);
                    clamp_rows(O1 == 1 ? 0 : w0 - L1 - r.g1, O1 == 1 ? 0 : w1 - L1 - r.g1 + (ACTIVE ? 1 : 0), O1, PADZ, gr0, gr1);
                }
            }
            const Cover cx = row_cover<ES>(pl, p.XP, S2, xr0, xr1), cg = row_cover<ES>(pl, p.OP, O2, gr0, gr1);
            stage_wave0(p.x, cx.first, cx.n, smem, XCOV + usedx);
            stage_wave0(p.go, cg.first, cg.n, smem, GCOV + usedg);
            r.xb = XCOV + usedx + cx.base;
            r.gb = GCOV + usedg + cg.base;
            r.xr0 = xr0;
            r.xr1 = xr1;
            r.gr0 = gr0;
            r.gr1 = gr1;
            usedx += cx.n * 16;
            usedg += cg.n * 16;
            if (tid == k) table[k] = r;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t f = f0 + static_cast<uint32_t>(tid * E);
    const bool mine = f < p.total;
    CT sums[4] = {CT(0), CT(0), CT(0), CT(0)};
    if (mine) {
        const uint32_t pl = fdiv(f, p.d_SP);
        const uint32_t r = f - pl * p.XP;
        int i = static_cast<int>(fdiv(r, p.d_SR));
        int j = static_cast<int>(r) - i * S2;
        const int slot = static_cast<int>(pl - plA);
        const Rec D0 = table[slot], D1 = table[min(slot + 1, nplanes - 1)];
        const int ecut = static_cast<int>(min(static_cast<uint32_t>(E), p.XP - r));
        const S *xg = static_cast<const S *>(p.x), *gg = static_cast<const S *>(p.go);
        const S zero = static_cast<S>(0.0f);
        const int lim = p.lds_bytes - ((S1 == 1 ? 0 : S2) + 2) * ES, glim = p.lds_bytes - ((O1 == 1 ? 0 : O2) + 2) * ES;   // (see flat_forward)
        const bool one_d = p.nd == 1;
        // the gradient at the output positions themselves.  Small planes: from the staged planes; large planes: from memory --
        // without a crop grad_out has grad_x's own geometry, the chunk is one aligned 16-byte load
        Chunk<S, E> own;
        bool own_loaded = false;
        if constexpr (!SMALL) {
            if (!cropped && f + E <= p.total) {
                own = load_chunk<S, E>(gg + f);
                own_loaded = true;
            }
        }
        Chunk<S, E> res;
        auto elements = [&](auto simple_tag) {   // (SIMPLE: see flat_forward)
        constexpr bool SIMPLE = decltype(simple_tag)::value;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool first = SIMPLE || e < ecut;
            const int c1 = first ? D0.c1 : D1.c1, c2 = first ? D0.c2 : D1.c2, g1 = first ? D0.g1 : D1.g1, g2 = first ? D0.g2 : D1.g2;
            const int xb = first ? D0.xb : D1.xb, gb = first ? D0.gb : D1.gb;
            const uint32_t ple = first ? pl : pl + 1u;
            const int io = i - L1, jo = j - L2;   // the element in grad_out's coordinates
            const bool inside = static_cast<unsigned>(io) < static_cast<unsigned>(O1) && static_cast<unsigned>(jo) < static_cast<unsigned>(O2) && f + e < p.total;
            // the gradient at the position itself (shifts_kernels.h:271)
            S gs;
            if constexpr (SMALL) {
                gs = lds_at<S>(smem, inside ? gb + (io * O2 + jo) * ES : kZero);
            } else {
                if (own_loaded) gs = inside ? own.e[e] : zero;
                else gs = inside ? gg[static_cast<uint64_t>(ple) * p.OP + static_cast<uint32_t>(io * O2 + jo)] : zero;
            }
            const CT g = widen<T>(gs);
            // the input corners around (i, j) - shift (:274-279): unfolded coordinates a, a + 1 x b, b + 1
            const int a = S1 == 1 ? 0 : i - c1, b = S2 == 1 ? 0 : j - c2;
            const int ga = O1 == 1 ? 0 : io - g1, gbc = O2 == 1 ? 0 : jo - g2;   // the gradient tap(s) of grad_x (:299-324)
            CT v0, v1, v2, v3;   // corner order: bit 0 = + 1 row, bit 1 = + 1 column
            if constexpr (PADZ) {
                // two addresses (rows a, a + 1; the columns b, b + 1 are adjacent), read unconditionally at a clamped address, masked
                // afterwards (elements outside the window read nothing: every mask carries `inside`)
                const bool ra = inside && static_cast<unsigned>(a) < static_cast<unsigned>(S1), ra1 = S1 == 1 ? ra : inside && static_cast<unsigned>(a + 1) < static_cast<unsigned>(S1);
                const bool cb = static_cast<unsigned>(b) < static_cast<unsigned>(S2), cb1 = S2 == 1 ? cb : static_cast<unsigned>(b + 1) < static_cast<unsigned>(S2);
                const int base = min(max(xb + (a * S2 + b) * ES, 0), lim);
                const int dn = S1 == 1 ? 0 : S2 * ES, rt = S2 == 1 ? 0 : ES;
                const S q00 = lds_at<S>(smem, base), q01 = lds_at<S>(smem, base + rt), q10 = lds_at<S>(smem, base + dn), q11 = lds_at<S>(smem, base + dn + rt);
                v0 = (ra && cb) ? widen<T>(q00) : CT(0);
                v1 = (ra1 && cb) ? widen<T>(q10) : CT(0);
                v2 = (ra && cb1) ? widen<T>(q01) : CT(0);
                v3 = (ra1 && cb1) ? widen<T>(q11) : CT(0);
            } else {
                const int xr0 = first ? D0.xr0 : D1.xr0, xr1 = first ? D0.xr1 : D1.xr1;
                auto xtap = [&](int ar, int bc) -> CT {   // folded coordinates: always a source element; outside the window: the zero words
                    if constexpr (SMALL || PAD == 1 || SIMPLE) {
                        return widen<T>(lds_at<S>(smem, inside ? xb + (ar * S2 + bc) * ES : kZero));
                    } else {
                        if (!inside) return CT(0);
                        if (ar >= xr0 && ar <= xr1) return widen<T>(lds_at<S>(smem, xb + (ar * S2 + bc) * ES));
                        return widen<T>(xg[static_cast<uint64_t>(ple) * p.XP + static_cast<uint32_t>(ar * S2 + bc)]);
                    }
                };
                constexpr bool ROWS_IN = SIMPLE && !SMALL && PAD >= 2;   // (the chunk's rows lie inside the plane / the window unfolded)
                const int ar = S1 == 1 ? 0 : (ROWS_IN ? a : fold_t<PAD>(a, S1, fS1)), ar1 = S1 == 1 ? 0 : (ROWS_IN ? a + 1 : fold_t<PAD>(a + 1, S1, fS1));
                const int bc = S2 == 1 ? 0 : fold_t<PAD>(b, S2, fS2), bc1 = S2 == 1 ? 0 : fold_t<PAD>(b + 1, S2, fS2);
                v0 = xtap(ar, bc);
                v1 = xtap(ar1, bc);
                v2 = xtap(ar, bc1);
                v3 = xtap(ar1, bc1);
            }
            const CT dA = v2 - v0, dB = v3 - v1;
            // (the product goes to the sums of the chunk's first plane or of its second; selected, not multiplied by zero: 0 x inf)
            const CT pA = g * dA, pB = g * dB;
            sums[0] += first ? pA : CT(0);
            sums[1] += first ? pB : CT(0);
            sums[2] += first ? CT(0) : pA;
            sums[3] += first ? CT(0) : pB;
            // grad_x (:299-324)
            if constexpr (PADZ) {
                const bool ra = inside && static_cast<unsigned>(ga) < static_cast<unsigned>(O1), cb = static_cast<unsigned>(gbc) < static_cast<unsigned>(O2);
                if constexpr (!ACTIVE) {
                    if constexpr (SMALL) {
                        res.e[e] = lds_at<S>(smem, (ra && cb) ? gb + (ga * O2 + gbc) * ES : kZero);
                    } else {
                        res.e[e] = lds_at<S>(smem, (ra && cb) ? gb + (ga * O2 + gbc) * ES : kZero);
                    }
                } else {
                    const bool ra1 = O1 == 1 ? ra : inside && static_cast<unsigned>(ga + 1) < static_cast<unsigned>(O1);
                    const bool cb1 = O2 == 1 ? cb : static_cast<unsigned>(gbc + 1) < static_cast<unsigned>(O2);
                    const int base = min(max(gb + (ga * O2 + gbc) * ES, 0), glim);
                    const int dn = O1 == 1 ? 0 : O2 * ES, rt = O2 == 1 ? 0 : ES;
                    const S q00 = lds_at<S>(smem, base), q01 = lds_at<S>(smem, base + rt), q10 = lds_at<S>(smem, base + dn), q11 = lds_at<S>(smem, base + dn + rt);
                    const CT fr[2] = {first ? D0.f1 : D1.f1, first ? D0.f2 : D1.f2};
                    const CT u00 = (ra && cb) ? widen<T>(q00) : CT(0), u10 = (ra1 && cb) ? widen<T>(q10) : CT(0);
                    const CT u01 = (ra && cb1) ? widen<T>(q01) : CT(0), u11 = (ra1 && cb1) ? widen<T>(q11) : CT(0);
                    S rv;
                    if (one_d) {
                        const CT u[2] = {u00, u01};
                        rv = narrow<T>(interp_t<T, 1>(u, fr + 1));
                    } else {
                        const CT u[4] = {u00, u10, u01, u11};
                        rv = narrow<T>(interp_t<T, 2>(u, fr));
                    }
                    res.e[e] = inside ? rv : zero;
                }
            } else {
                const int gr0 = first ? D0.gr0 : D1.gr0, gr1 = first ? D0.gr1 : D1.gr1;
                auto gtap = [&](int ar, int bc) -> S {   // grad_out (row ar, column bc), folded in the window's sizes (:295-297, :319-324)
                    if constexpr (SMALL || PAD == 1 || SIMPLE) {
                        return lds_at<S>(smem, inside ? gb + (ar * O2 + bc) * ES : kZero);
                    } else {
                        if (!inside) return zero;
                        if (ar >= gr0 && ar <= gr1) return lds_at<S>(smem, gb + (ar * O2 + bc) * ES);
                        return gg[static_cast<uint64_t>(ple) * p.OP + static_cast<uint32_t>(ar * O2 + bc)];
                    }
                };
                constexpr bool GROWS_IN = SIMPLE && !SMALL && PAD >= 2;
                const int ar = O1 == 1 ? 0 : (GROWS_IN ? ga : fold_t<PAD>(ga, O1, fO1)), bc = O2 == 1 ? 0 : fold_t<PAD>(gbc, O2, fO2);
                if constexpr (!ACTIVE) {
                    res.e[e] = gtap(ar, bc);
                } else {
                    const int ar1 = O1 == 1 ? 0 : (GROWS_IN ? ga + 1 : fold_t<PAD>(ga + 1, O1, fO1)), bc1 = O2 == 1 ? 0 : fold_t<PAD>(gbc + 1, O2, fO2);
                    const CT fr[2] = {first ? D0.f1 : D1.f1, first ? D0.f2 : D1.f2};
                    S rv;
                    if (one_d) {
                        const CT u[2] = {widen<T>(gtap(ar, bc)), widen<T>(gtap(ar, bc1))};
                        rv = narrow<T>(interp_t<T, 1>(u, fr + 1));
                    } else {
                        const CT u[4] = {widen<T>(gtap(ar, bc)), widen<T>(gtap(ar1, bc)), widen<T>(gtap(ar, bc1)), widen<T>(gtap(ar1, bc1))};
                        rv = narrow<T>(interp_t<T, 2>(u, fr));
                    }
                    res.e[e] = inside ? rv : zero;
                }
            }
            ++j;
            if (j == S2) {
                j = 0;
                ++i;
                if (i == S1) i = 0;
            }
        }
        };
        bool simple = ecut == E;
        if constexpr (!SMALL && PAD >= 2) {
            // rows this chunk reads, unfolded, of the input and of the gradient: inside the plane / the window nothing folds
            const int ilast = static_cast<int>(fdiv(static_cast<uint32_t>(r) + E - 1, p.d_SR));
            const int a_lo = i - D0.c1, a_hi = ilast - D0.c1 + 1, g_lo = i - L1 - D0.g1, g_hi = ilast - L1 - D0.g1 + (ACTIVE ? 1 : 0);
            simple = simple && (S1 == 1 || (a_lo >= 0 && a_hi < S1)) && (O1 == 1 || (g_lo >= 0 && g_hi < O1));
        }
        if (__all(simple)) elements(std::true_type{});
        else elements(std::false_type{});
        S *op = static_cast<S *>(p.out) + f;
        if (f + E <= p.total) {
            store_chunk<S, E>(op, res);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (f + e < p.total) op[e] = res.e[e];
        }
    }
    // ---- the step's weight-gradient records: one per plane it touches, the chunks' sums added in chunk order -----------------
    CT *part = reinterpret_cast<CT *>(smem + PART);
#pragma unroll
    for (int k = 0; k < 4; ++k) part[tid * 4 + k] = sums[k];
    __syncthreads();
    const int grp = tid >> 4, lane = tid & 15;
    for (int k = grp; k < nplanes; k += kThreads / 16) {
        const uint32_t pl = plA + static_cast<uint32_t>(k);
        // the plane's elements inside the step, step-local
        const uint32_t lo = pl * p.XP > f0 ? pl * p.XP - f0 : 0u;
        const uint32_t hi = min(f1 - f0, (pl + 1u) * p.XP - 1u - f0);
        const int tlo = static_cast<int>(lo / E), thi = static_cast<int>(hi / E);
        double a = 0.0, b = 0.0;
        for (int t = tlo + lane; t <= thi; t += 16) {
            // chunk t holds this plane as its first plane unless the plane starts inside it
            const int sl = (static_cast<uint32_t>(t) * E < lo) ? 2 : 0;
            a += static_cast<double>(part[t * 4 + sl]);
            b += static_cast<double>(part[t * 4 + sl + 1]);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {   // fixed tree over the 16 lanes
            a += __shfl_down(a, off, 16);
            b += __shfl_down(b, off, 16);
        }
        if (lane == 0) {
            double *rec = p.partials + (static_cast<size_t>(bid) + pl) * 2;
            rec[0] = a;
            rec[1] = b;
        }
    }
}

// grad_w[c][:] = blend(sum over the records of channel c's planes, in a fixed order): one wave per channel, lanes over the batch
template <typename T>
__global__ __launch_bounds__(64) void flat_reduce(const FlatParams p, int N, int active, typename T::S *__restrict__ grad_w) {
    using S = typename T::S;
    constexpr int E = 16 / sizeof(S);
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int n = threadIdx.x; n < N; n += 64) {
        const uint64_t pl = static_cast<uint64_t>(n) * p.C + c;
        const uint64_t e0 = pl * p.XP, e1 = e0 + p.XP - 1;
        const uint64_t s0 = e0 / (kThreads * E), s1 = e1 / (kThreads * E);
        for (uint64_t s = s0; s <= s1; ++s) {
            const double *rec = p.partials + (s + pl) * 2;
            a += rec[0];
            b += rec[1];
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        double out[2];
        if (p.nd == 1) {
            out[0] = a;   // 1-D: the one partial is the column difference (interp1D_dx, interpolation.h:10-13)
            out[1] = 0.0;
        } else {
            using CT = typename T::C;
            int u0, u1, u2, u3;
            CT f1, f2;
            if (active) flat_channel_backward<CT, true>(p, c, u0, u1, u2, u3, f1, f2);
            else flat_channel_backward<CT, false>(p, c, u0, u1, u2, u3, f1, f2);
            const double s[2] = {a, b}, dwd[2] = {static_cast<double>(f1), static_cast<double>(f2)};   // (exactly as the compute type holds them)
            blend_diffs<2>(s, dwd, out);
        }
        for (int k = 0; k < p.nd; ++k) {
            if constexpr (sizeof(S) == 8) grad_w[c * p.nd + k] = out[k];
            else grad_w[c * p.nd + k] = narrow<T>(static_cast<float>(out[k]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct FlatPlan {
    bool ok, small;
    size_t lds;        // dynamic LDS bytes
    int gcov_off;      // backward: offset of the gradient's cover behind the x cover
    int front;         // slack in front of the first cover
    uint64_t steps;
};

// ceil-ish bound of the cover of `planes_max` whole planes / of (rows + extra) rows, in bytes, 16-byte pieces
inline int64_t cover_bytes(int64_t payload) { return ((payload + 15 + 15) / 16) * 16; }

// backward = true: the streamed tensor has x's geometry (S), staged: x (S) and grad_out (O); forward: streamed out (O), staged x (S)
FlatPlan flat_plan(const Geometry &g, int es, bool backward) {
    FlatPlan pl{};
    const int E = 16 / es;
    const int64_t S1 = g.S[1], S2 = g.S[2], O1 = g.O[1], O2 = g.O[2];
    const int64_t XPB = S1 * S2 * es, OPB = O1 * O2 * es;
    const int64_t SPB = backward ? XPB : OPB;          // plane bytes of the streamed tensor
    const int64_t total = g.N * g.C * (backward ? S1 * S2 : O1 * O2);
    pl.steps = static_cast<uint64_t>((total + kThreads * E - 1) / (kThreads * E));
    const int64_t step_bytes = kThreads * 16;
    const int64_t nplanes = (step_bytes + SPB - 1) / SPB + 1;                 // planes a step can touch
    const int rec = es == 8 ? 56 : 48;   // sizeof(PlaneRec<CT>)
    // one row (+ 2 elements) of slack behind the covers: the clamp of the zeros-padding corner reads never moves a valid group
    const size_t slack = static_cast<size_t>(((std::max(S1 == 1 ? 0 : S2, O1 == 1 ? 0 : O2) + 2) * es + 15) / 16 * 16);
    // small: whole planes of every staged tensor
    const int64_t cx_small = cover_bytes(nplanes * XPB), cg_small = cover_bytes(nplanes * OPB);
    if (nplanes <= kMaxPlanes && cx_small <= kCoverBudget && (!backward || cg_small <= kCoverBudget)) {
        pl.ok = true;
        pl.small = true;
        const int64_t tab = kTab + ((kMaxPlanes * rec + 15) / 16) * 16;
        pl.gcov_off = static_cast<int>(cx_small);
        pl.front = static_cast<int>(slack);
        pl.lds = static_cast<size_t>(tab + (backward ? kThreads * 4 * (es == 8 ? 8 : 4) + cx_small + cg_small : cx_small)) + 2 * slack;
        //  the slack is a whole row: planes of two very wide ragged rows pass the cover budget and still exceed the
        // 64 KiB a launch may ask for -- decline, the router falls through to the next family
        if (pl.lds > 64 * 1024) pl.ok = false;
        return pl;
    }
    // large: at most two planes per step, row ranges
    if (SPB < step_bytes) return pl;
    const int64_t SR = backward ? S2 : O2;                                     // row length of the streamed tensor
    const int64_t rows = (kThreads * E + SR - 2) / SR + 1;                     // rows a step can touch (both planes together: + 1)
    // per plane (rows_k + 1 (+1)) source rows; two planes: rows + 1 + 2 * 2
    const int64_t cx_large = cover_bytes((rows + 5) * S2 * es) + 32, cg_large = cover_bytes((rows + 5) * O2 * es) + 32;
    if (cx_large > kCoverBudget || (backward && cg_large > kCoverBudget)) return pl;
    pl.ok = true;
    pl.small = false;
    const int64_t tab = kTab + ((2 * rec + 15) / 16) * 16;
    pl.gcov_off = static_cast<int>(cx_large);
    pl.front = static_cast<int>(slack);
    pl.lds = static_cast<size_t>(tab + (backward ? kThreads * 4 * (es == 8 ? 8 : 4) + cx_large + cg_large : cx_large)) + 2 * slack;
    if (pl.lds > 64 * 1024) pl.ok = false;
    return pl;
}

thread_local int g_flat_tune = 0;   // knob 27: 0 automatic, 1 never, 2 whenever eligible

bool flat_common_ok(const Geometry &g, int dtype, const void *a, const void *b) {
    if (g_flat_tune == 1) return false;
    if (dtype > SHIFTND_BF16 || (g.nd != 1 && g.nd != 2) || g.K[0] > 0) return false;
    if (g.S[0] != 1 || g.O[0] != 1 || g.S[1] < 1 || g.S[2] < 1 || g.O[1] < 1 || g.O[2] < 1 || g.N < 1 || g.C < 1) return false;
    if (g.nd == 1 && (g.S[1] != 1 || g.O[1] != 1)) return false;
    const int es = dtype_size(dtype);
    // 32-bit element and byte arithmetic in the kernels
    if (g.N * g.C * g.S[1] * g.S[2] * es >= (1LL << 32) || g.N * g.C * g.O[1] * g.O[2] * es >= (1LL << 32)) return false;
    if (g.N * g.C >= (1LL << 31) || g.S[1] >= (1 << 15) || g.S[2] >= (1 << 15) || g.S[1] * g.S[2] * es >= (1LL << 26)) return false;
    if (reinterpret_cast<uintptr_t>(a) % 16 || reinterpret_cast<uintptr_t>(b) % 16) return false;
    return true;
}

template <typename T, bool ACTIVE, bool SMALL>
void launch_flat_forward(const FlatParams &p, size_t lds, int pad, hipStream_t st) {
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    switch (pad) {
    case 0: hipLaunchKernelGGL((flat_forward<T, ACTIVE, 0, SMALL>), grid, block, lds, st, p); break;
    case 1: hipLaunchKernelGGL((flat_forward<T, ACTIVE, 1, SMALL>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((flat_forward<T, ACTIVE, kPadRT, SMALL>), grid, block, lds, st, p); break;   // periodic / reflect / symmetric: one instantiation, p.pad
    }
}

template <typename T, bool ACTIVE, bool SMALL>
void launch_flat_backward(const FlatParams &p, const FlatPlan &pl, int pad, int N, void *gw, hipStream_t st) {
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    if (!FLAT_INLINE_PREP) hipLaunchKernelGGL((flat_prep<T>), dim3((p.C + kThreads - 1) / kThreads), block, 0, st, p, ACTIVE ? 1 : 0);
    switch (pad) {
    case 0: hipLaunchKernelGGL((flat_backward<T, ACTIVE, 0, SMALL>), grid, block, pl.lds, st, p, pl.gcov_off); break;
    case 1: hipLaunchKernelGGL((flat_backward<T, ACTIVE, 1, SMALL>), grid, block, pl.lds, st, p, pl.gcov_off); break;
    default: hipLaunchKernelGGL((flat_backward<T, ACTIVE, kPadRT, SMALL>), grid, block, pl.lds, st, p, pl.gcov_off); break;   // periodic / reflect / symmetric: one instantiation, p.pad
    }
    hipLaunchKernelGGL((flat_reduce<T>), dim3(p.C), dim3(64), 0, st, p, N, ACTIVE ? 1 : 0, static_cast<typename T::S *>(gw));
}

void fill_params(FlatParams &p, const Geometry &g, int es, bool backward, const FlatPlan &pl) {
    p.C = static_cast<int>(g.C);
    p.pad = g.pad;
    p.nd = g.nd;
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.XP = static_cast<uint32_t>(g.S[1] * g.S[2]);
    p.OP = static_cast<uint32_t>(g.O[1] * g.O[2]);
    p.planes = static_cast<uint32_t>(g.N * g.C);
    p.total = p.planes * (backward ? p.XP : p.OP);
    p.total_steps = static_cast<uint32_t>(pl.steps);
    p.steps_per_xcd = (p.total_steps + 7) / 8;
    p.d_SP = make_fastdiv(backward ? p.XP : p.OP);
    p.d_SR = make_fastdiv(static_cast<uint32_t>(backward ? p.S2 : p.O2));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, p.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, p.pad)));
    p.d_pero1 = make_fastdiv(static_cast<uint32_t>(map_period(p.O1, p.pad)));
    p.d_pero2 = make_fastdiv(static_cast<uint32_t>(map_period(p.O2, p.pad)));
    p.lds_bytes = static_cast<int>(pl.lds);
    p.front = pl.front;
}

}  // namespace

void flat_set_tuning(int value) { g_flat_tune = value; }

// contiguous 1-D / 2-D float tensors (a window included), any row length; taken automatically where the chunk kernels are not
// eligible (rows or planes that are not whole 16-byte pieces)
bool flat_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (!flat_common_ok(g, dtype, x, out)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    return flat_plan(g, dtype_size(dtype), false).ok;
}

int flat_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const FlatPlan pl = flat_plan(g, es, false);
    FlatParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    fill_params(p, g, es, false, pl);
    const bool active = g.active != 0;
    note_kernel(active ? "flat_active_forward" : "flat_gather_forward");
    // (the sparse shift is a raw copy: one instantiation per element size)
#define SHIFTND_FLAT_FWD(TT, ACT) \
    if (pl.small) launch_flat_forward<TT, ACT, true>(p, pl.lds, g.pad, st); \
    else launch_flat_forward<TT, ACT, false>(p, pl.lds, g.pad, st);
    if (!active) {
        if (es == 2) { SHIFTND_FLAT_FWD(f16_t, false) } else if (es == 4) { SHIFTND_FLAT_FWD(f32_t, false) } else { SHIFTND_FLAT_FWD(f64_t, false) }
    } else if (dtype == SHIFTND_F32) { SHIFTND_FLAT_FWD(f32_t, true)
    } else if (dtype == SHIFTND_F64) { SHIFTND_FLAT_FWD(f64_t, true)
    } else if (dtype == SHIFTND_F16) { SHIFTND_FLAT_FWD(f16_t, true)
    } else { SHIFTND_FLAT_FWD(bf16_t, true) }
#undef SHIFTND_FLAT_FWD
    return SHIFTND_OK;
}

bool flat_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (!flat_common_ok(g, dtype, x, gx) || reinterpret_cast<uintptr_t>(go) % 16) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O) || !dense(g.gs, g.N, g.C, g.S)) return false;
    return flat_plan(g, dtype_size(dtype), true).ok;
}

// geometry only: records of the widest plan (2-byte elements: the fewest elements per step is the 8-byte type's)
size_t flat_backward_workspace(const Geometry &g) {
    if ((g.nd != 1 && g.nd != 2) || g.N < 1 || g.C < 1 || g.S[1] * g.S[2] < 1) return 0;
    if (g.N * g.C * g.S[1] * g.S[2] >= (1LL << 32)) return 0;
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * g.S[1] * g.S[2];
    const uint64_t steps = (total + kThreads * 2 - 1) / (kThreads * 2);   // E = 2 (fp64): the most steps
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    return up((steps + static_cast<uint64_t>(g.N) * g.C + 1) * 2 * sizeof(double)) + up(static_cast<size_t>(g.C) * sizeof(FlatDesc));
}

int flat_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                  hipStream_t st) {
    const int es = dtype_size(dtype);
    const FlatPlan pl = flat_plan(g, es, true);
    FlatParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    p.wkind = dtype;
    fill_params(p, g, es, true, pl);
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    p.partials = static_cast<double *>(workspace);
    p.desc = reinterpret_cast<FlatDesc *>(static_cast<char *>(workspace) + up((pl.steps + static_cast<uint64_t>(g.N) * g.C + 1) * 2 * sizeof(double)));
    const bool active = g.active != 0;
    note_kernel("flat_backward");
#define SHIFTND_FLAT_BWD(TT, ACT) \
    if (pl.small) launch_flat_backward<TT, ACT, true>(p, pl, g.pad, static_cast<int>(g.N), gw, st); \
    else launch_flat_backward<TT, ACT, false>(p, pl, g.pad, static_cast<int>(g.N), gw, st);
#define SHIFTND_FLAT_BWD2(TT) \
    if (active) { SHIFTND_FLAT_BWD(TT, true) } else { SHIFTND_FLAT_BWD(TT, false) }
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_FLAT_BWD2(f32_t) break;
    case SHIFTND_F64: SHIFTND_FLAT_BWD2(f64_t) break;
    case SHIFTND_F16: SHIFTND_FLAT_BWD2(f16_t) break;
    default: SHIFTND_FLAT_BWD2(bf16_t) break;
    }
#undef SHIFTND_FLAT_BWD2
#undef SHIFTND_FLAT_BWD
    return SHIFTND_OK;
}

}  // namespace shiftnd
