// shiftnd_walk3.hip -- round 3's walk through the planes (split from shiftnd_step.hip in round 4): the 3-D
// interpolating forward and both 3-D backwards of fp32 / fp64 tensors in the reference's nesting (bit-exact), and the fused
// average-pool variants of every float dtype.  16-bit tensors without a pool: shiftnd_walk.hip (round 4).
// Reference behaviour restated: kernels/shifts_kernels.h:156-220, :222-327, :132-154; kernels/interpolation.h:34-61.
#include "shiftnd_step.hpp"

namespace shiftnd {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// walk_forward: the 3-D interpolating forward as a walk along dim0.  step_forward_lds<T, 3> reads, masks and widens four
// corner rows per output row (C3, bf16: 220 vector instructions per 8-element chunk, 82 % VALU-busy at 3.8 TB/s); the
// sliding-window kernel (shiftnd_slide.hip) carries half of them in registers but walks down the ROWS of 16 planes at
// once: 224-byte pieces 25 KB apart, 1024 long workgroups.  Here a workgroup owns R consecutive rows (a contiguous
// R x row-bytes run of every plane) of one (n, c) volume and walks through its planes a = 0 .. O0 - 1: source plane
// map(a + 1) of step a IS source plane map(a) of step a + 1 -- for every padding, the map is the same expression -- so
// the two corner rows of the "+1" plane stay in registers (widened) and become the "+0" plane's rows of the next step.
// Per step: ONE plane's R + 1 rows staged (global_load_lds, every thread its own piece: the row and column maps of a
// thread never change along the walk), two row windows read and widened instead of four, the blends nested as the
// reference nests them (plane, row, inner: interpolation.h:34-40; same bits as interp_nd).  Contiguous 3-D tensors
// without crop; every float dtype.
// ---------------------------------------------------------------------------------------------------------------------
// POOL: the module's average pool (windows (K0, K1, 2), K1 <= 2, ceil mode) as the epilogue: a plane's values are rounded to the
// storage type like the two-step sequence's shift output and summed in ATen's order -- plane, row, column -- in the compute
// type: the two columns of a window sit in one thread, its planes arrive in consecutive steps (the sum stays in registers),
// and the second row belongs to the thread one row down, which leaves its chunk in LDS for the next step (two alternating
// slots: no extra barrier).  Only the pooled tensor is written.
template <typename T, int PAD, bool POOL = false>
__global__ __launch_bounds__(kThreads) void walk_forward(const FwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;  // 64-byte pads in front and behind: see lds_read_row

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wv[3];
    load_weights_nd<CT>(p.w, p.wkind, c, p.nd, wv);
    CT rr[3], dn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rr[d] = c_floor<CT>(wv[d]);
        dn[d] = wv[d] - rr[d];
    }
    const int cs0 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], p.S0, p.d_per0, p.pad));
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], p.S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], p.S2, p.d_per2, p.pad));
    const CT dw[3] = {dn[0], dn[1], dn[2]};

    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr;
    const int b0 = step * R;
    const int Rn = min(R, p.O1 - b0);
    const char *xp = reinterpret_cast<const char *>(static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane);
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * (POOL ? p.p_plane : p.o_plane);

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    // the piece this thread stages, the same for every plane: piece tc of row tr, rows 0 .. min(R, Rn) -- the host picks R with
    // (R + 1) * cpr <= 256, so the + 1 corner row of the step's last row has its threads too.  Staging goes global -> registers
    // -> LDS, two planes ahead (two registers sets alternate, the loop unrolled by two; see walk_backward); a thread without a
    // piece, a fill row and a fill plane load zeros (out-of-range offset / empty resource) and park them.
    (void)wave;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const bool own = tr <= R && tr <= Rn;
    const int src_own = own ? row_map_t<PAD>(b0 + tr, cs1, S1, p.pad) : -1;
    const uint32_t plane_bytes = static_cast<uint32_t>(S1) * static_cast<uint32_t>(S2) * static_cast<uint32_t>(sizeof(S));
    const uint32_t voff = src_own >= 0 ? static_cast<uint32_t>(src_own * S2 + tc * E) * static_cast<uint32_t>(sizeof(S)) : 0x80000000u;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, static_cast<uint32_t>(S0) * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, 0, 0x00020000);
    char *park_at = own ? tile + tid * 16 : tile + (R + 1) * cpr * 16 + 64 + (POOL ? 2 * kThreads * 16 + 64 : 0) + tid * 16;   // (a private dump slot)
    auto load_plane = [&](int pa) {   // source plane pa (uniform; -1: fill)
        return __builtin_amdgcn_raw_buffer_load_b128(pa >= 0 ? xres : none, voff, pa >= 0 ? static_cast<uint32_t>(pa) * plane_bytes : 0u, 0);
    };
    auto park = [&](const u4 &v) { *reinterpret_cast<u4 *>(__builtin_assume_aligned(park_at, 16)) = v; };
    const int jo = tc * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int base = jo - cs2;
        xm.base = (base + E < 0 || base >= S2) ? 0 : base;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = (base + e >= 0 && base + e < S2) ? base + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(jo, cs2, S2, p.pad);
    }
    // window reads: two aligned 16-byte spans and the workgroup's phase (lds_window6: no bank conflicts)
    const int phw = (-cs2 * static_cast<int>(sizeof(S))) & 15;
    const bool fastw = xm.affine && (PAD == 0 || ((xm.base * static_cast<int>(sizeof(S))) & 15) == phw);
    const bool mine = tr < R && tr < Rn;   // this thread produces a chunk
    const int b = b0 + tr;
    bool rv[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) rv[hb] = PAD != 0 || row_map_t<PAD>(b + hb, cs1, S1, p.pad) >= 0;
    const int RBL = cpr * 16;  // bytes per staged row
    const char *rows = tile + tr * RBL;

    // plane map(0): the first step's "+0" rows
    CT carried[2][E + 1];
    {
        const int pa0 = row_map_t<PAD>(0, cs0, S0, p.pad);
        park(load_plane(pa0));
        __syncthreads();
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            S r0[E + 1];
            lds_read_row_span<S, E>(rows + hb * RBL, mine && rv[hb] && pa0 >= 0, xm, fastw, phw, r0);
#pragma unroll
            for (int e = 0; e <= E; ++e) carried[hb][e] = widen<T>(r0[e]);
        }
    }
    const int64_t out_plane = static_cast<int64_t>(p.O1) * p.O2;
    S *orow = op + static_cast<int64_t>(b) * p.O2 + jo;
    // POOL state: the thread that owns the first row of a window (every thread when K1 == 1) accumulates its E / 2 windows
    char *xch = tile + (R + 1) * RBL + 64;                         // [2][kThreads] chunks: the second rows of the windows
    const bool pairs = POOL && p.K1 == 2;
    const bool first_row = POOL && mine && (!pairs || (tr & 1) == 0);
    const int n1 = pairs ? min(2, p.O1 - b) : 1;                  // rows of this thread's windows (a ragged last row: 1)
    const int pr = pairs ? (b >> 1) : b;
    CT pacc[E / 2 > 0 ? E / 2 : 1];
    Chunk<S, E> prev;                                             // this thread's chunk of the previous plane
    auto pool_plane = [&](int ap) {   // fold plane ap (own chunk `prev`, the row below from the exchange slot) into the windows
        const int pp = static_cast<int>(fdiv(static_cast<uint32_t>(ap), p.d_k0));
        const int u0 = ap - pp * p.K0, n0 = min(p.K0, p.O0 - pp * p.K0);
        if (u0 == 0) {
#pragma unroll
            for (int j = 0; j < E / 2; ++j) pacc[j] = CT(0);
        }
#pragma unroll
        for (int j = 0; j < E / 2; ++j) pacc[j] = (pacc[j] + widen<T>(prev.e[2 * j])) + widen<T>(prev.e[2 * j + 1]);
        if (n1 == 2) {
            Chunk<S, E> below;
            __builtin_memcpy(below.e, __builtin_assume_aligned(xch + ((ap & 1) * kThreads + tid) * 16, 16), 16);
#pragma unroll
            for (int j = 0; j < E / 2; ++j) pacc[j] = (pacc[j] + widen<T>(below.e[2 * j])) + widen<T>(below.e[2 * j + 1]);
        }
        if (u0 == n0 - 1) {
            Chunk<S, (E / 2 > 0 ? E / 2 : 1)> outc;
#pragma unroll
            for (int j = 0; j < E / 2; ++j) outc.e[j] = narrow<T>(div_count<CT>(pacc[j], n0 * n1 * 2));
            S *dst = op + (static_cast<int64_t>(pp) * p.P1 + pr) * p.P2 + jo / 2;
            __builtin_memcpy(__builtin_assume_aligned(dst, sizeof(S) * E / 2), outc.e, sizeof(S) * (E / 2));
        }
    };
    __syncthreads();   // the first plane has been read
    u4 stA = load_plane(row_map_t<PAD>(1, cs0, S0, p.pad));                            // the "+1" plane of step 0
    u4 stB = load_plane(1 < p.O0 ? row_map_t<PAD>(2, cs0, S0, p.pad) : -1);            // ... of step 1
    auto walk_step = [&](int a, u4 &pend) {   // `pend`: the "+1" plane of step a; leaves with that of step a + 2 in flight
        const int pa1 = row_map_t<PAD>(a + 1, cs0, S0, p.pad);
        park(pend);
        __syncthreads();
        pend = load_plane(a + 2 < p.O0 ? row_map_t<PAD>(a + 3, cs0, S0, p.pad) : -1);
        CT rowb[2][E + 1];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            S r1[E + 1];
            lds_read_row_span<S, E>(rows + hb * RBL, mine && rv[hb] && pa1 >= 0, xm, fastw, phw, r1);
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                const CT nv = widen<T>(r1[e]);
                const CT two[2] = {carried[hb][e], nv};
                rowb[hb][e] = interp_t<T, 1>(two, &dw[0]);
                carried[hb][e] = nv;
            }
        }
        CT colb[E + 1];
#pragma unroll
        for (int e = 0; e <= E; ++e) {
            const CT two[2] = {rowb[0][e], rowb[1][e]};
            colb[e] = interp_t<T, 1>(two, &dw[1]);
        }
        Chunk<S, E> res;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const CT two[2] = {colb[e], colb[e + 1]};
            res.e[e] = narrow<T>(interp_t<T, 1>(two, &dw[2]));
        }
        if constexpr (POOL) {
            if (first_row && a > 0) pool_plane(a - 1);   // (the row below left its chunk of plane a - 1 before this step's barriers)
            prev = res;
            if (pairs && mine && (tr & 1)) __builtin_memcpy(__builtin_assume_aligned(xch + ((a & 1) * kThreads + tid - cpr) * 16, 16), res.e, 16);
        } else {
            if (mine) store_chunk<S, E>(orow + a * out_plane, res);
        }
        __syncthreads();   // everybody has read this step's plane
    };
    int a = 0;
    for (; a + 1 < p.O0; a += 2) {   // whole pairs: no condition between the steps
        walk_step(a, stA);
        walk_step(a + 1, stB);
    }
    if (a < p.O0) walk_step(a, stA);
    if constexpr (POOL) {
        __syncthreads();
        if (first_row) pool_plane(p.O0 - 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// walk_backward: the 3-D interpolating backward as the same walk along dim0 (see walk_forward).  A workgroup owns R rows of
// one (n, c) volume and walks through its planes; per step it stages ONE plane of the saved input and ONE of the incoming
// gradient (R + 1 rows each: the "+1" corner planes map0(a + 1) of the two maps) and reads the thread's own gradient chunk
// straight from memory; the "+0" corner planes are the previous step's "+1" planes, widened, in registers.  grad_x: the
// blends nested as the reference nests them (plane, row, inner) -- 3 E + 3 instead of 7 E, same bits as interp_nd; the
// weight gradient: the eight corner-difference sums of step_backward<T, 3>, accumulated over the walk (fp32 per step,
// folded into fp64 every four planes), one record per workgroup for step_reduce.
// ---------------------------------------------------------------------------------------------------------------------
// POOL: `go` is the gradient of the POOLED output [N, C, P0, P1, P2] (window = stride = (K0, K1, 2)): every 16-byte piece of the
// unpooled gradient the walk consumes -- the staged corner rows and the thread's own chunk -- is 8 bytes of a pooled row, loaded
// as they are and expanded when they are parked / used: g = pooled / (window size), rounded to the storage type like the two-step
// sequence (ATen's avg_pool backward).  The rest of the kernel does not know.
// ACTIVE = false: the sparse shift.  Its weight gradient is the same eight corner sums (the x corners around i - round(w),
// fractions frac(|w|): shifts_cpu.cpp:242-244); its grad_x is ONE tap of the gradient -- go(g0[a], g1[b], gcol[j]) -- so the
// step stages the gradient plane g0[a] itself (no "+1" plane, nothing carried) and copies the window.
// CROP (round 6; zeros padding, no pool): `go` is the gradient of a WINDOW [wO0, wO1, wO2] that begins at (wL0, wL1, wL2) of the volume
// (ops/shifts.cpp:93-135) -- see walk_backward16<.., CROP> (shiftnd_walk.hip): the staged gradient rows are window rows as they lie
// from window column 0, the crop along the row is one more column shift of the gradient's map (cg2 + wL2, columns beyond wO2 masked by
// the map), rows and planes are offsets of the staged index, the own chunk comes from the two aligned pieces around it, and grad_x is
// zero outside the window.
template <typename T, int PAD, bool POOL = false, bool ACTIVE = true, bool CROP = false>
__global__ __launch_bounds__(kThreads) void walk_backward(const StepParams p) {
    static_assert(!CROP || PAD == 0, "the cropped walk: zeros padding");
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int REC = RecSize<E>::N;
    constexpr int NDIFF = WDiff<3>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spv);   // (n, c); its spv = (depth parts) * (row steps) workgroups
    const uint32_t vrem = bid - plane * static_cast<uint32_t>(p.spv);
    const int dq = static_cast<int>(fdiv(vrem, p.d_spp));
    const int step = static_cast<int>(vrem) - dq * p.spp;
    const int a0 = dq * p.walk_planes, a1 = min(p.S0, a0 + p.walk_planes);   // the planes this workgroup walks through
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr;
    const int b0 = step * R;
    const int Rn = min(R, S1 - b0);
    const int RB = S2 * static_cast<int>(sizeof(S));
    const char *xp = reinterpret_cast<const char *>(static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane);
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * ((POOL || CROP) ? p.g_plane : p.x_plane);
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane;
    // the gradient's geometry: the volume's, or (CROP) the window's
    const int O0 = CROP ? p.wO0 : S0, O1 = CROP ? p.wO1 : S1, O2 = CROP ? p.wO2 : S2;
    const int L0 = CROP ? p.wL0 : 0, L1 = CROP ? p.wL1 : 0, L2 = CROP ? p.wL2 : 0;

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    const int ji = tc * E;
    ColState<E> xm, gm;
    if constexpr (PAD == 0) {
        auto affine_state = [&](int cs, int len) {
            ColState<E> st;
            st.base = ji - cs;
            if (st.base + E < 0 || st.base >= len) st.base = 0;
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) st.cm[e] = (ji - cs + e >= 0 && ji - cs + e < len) ? ji - cs + e : -1;
            return st;
        };
        xm = affine_state(d.cx2, S2);
        gm = affine_state(d.cg2 + L2, O2);
    } else {
        const size_t rec = (static_cast<size_t>(c) * cpr + tc) * REC;
        xm = load_colstate<E>(p.colx + rec);
        gm = load_colstate<E>(p.colg + rec);
    }
    // the pieces this thread stages, the same for every plane (see walk_forward)
    // (the host picks R with (R + 1) * cpr <= 256: thread (tr, tc), tr <= R, stages piece tc of row tr -- the "+1" corner row of
    // the step's last row included -- so a plane costs one load per tensor and thread, and two planes can be in flight)
    const bool own = tr <= R && tr <= Rn;
    const int sx_own = own ? row_map_t<PAD>(b0 + tr, d.cx1, S1, p.pad) : -1;
    const int sg_own = (own && (ACTIVE || tr < R)) ? row_map_t<PAD>(b0 + tr - L1, d.cg1, O1, p.pad) : -1;
    auto piece_off = [&](int row, int piece, int rowlen) { return static_cast<uint32_t>(max(row, 0) * rowlen + piece * E) * static_cast<uint32_t>(sizeof(S)); };
    const uint32_t ox_own = piece_off(sx_own, tc, S2), og_own = piece_off(sg_own, tc, O2);
    const uint32_t plane_bytes = static_cast<uint32_t>(S1) * static_cast<uint32_t>(S2) * static_cast<uint32_t>(sizeof(S));
    const uint32_t gplane_bytes = CROP ? static_cast<uint32_t>(O1) * static_cast<uint32_t>(O2) * static_cast<uint32_t>(sizeof(S)) : plane_bytes;
    const int GP0 = (R + 1) * cpr;   // first LDS piece of the gradient group
    // Staging goes global -> registers -> LDS, TWO planes ahead (two register sets alternate, the loop is unrolled by two): the
    // loads of planes a + 2 and a + 3 are in flight while step a is computed -- a step lasts about as long as a memory round
    // trip under load, one plane ahead left the parking store waiting (an LDS-DMA in flight would make hipcc wait for it
    // before the first LDS read of the compute phase).  Every
    // memory instruction of the loop is unconditional -- a thread without a piece (or a plane that is fill) uses an
    // out-of-range buffer offset / an empty resource, which loads zeros, and parks them in its private dump slot -- so the
    // compiler's wait counts are exact.
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t kOOR = 0x80000000u;
    constexpr int kRsrcFlags = 0x00020000;
    const uint32_t vol_bytes = static_cast<uint32_t>(S0) * plane_bytes;   // < 2^31 (host)
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, vol_bytes, kRsrcFlags);
    // POOL + CROP of 16-bit elements: the pooled rows of a window (110 columns -> 55 pooled elements) start at 2-byte boundaries, and so may the
    // (n, c) volume of the pooled gradient; buffer loads want dwords.  The resource starts at the dword below the volume and every 8
    // pooled bytes are taken from the three dwords around them (pooled8)
    constexpr bool ODD16 = POOL && CROP && sizeof(S) == 2;
    const uint32_t gmis = ODD16 ? static_cast<uint32_t>(reinterpret_cast<uintptr_t>(gp) & 3u) : 0u;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(gp) - gmis), 0,
        ((POOL || CROP) ? static_cast<uint32_t>(p.g_plane) * static_cast<uint32_t>(sizeof(S)) : vol_bytes) + gmis, kRsrcFlags);
    // POOL: the 8 bytes of pooled row `row / K1` under piece `piece` of unpooled row `row`, bytes within a pooled plane; the
    // window rows the pooled row averages
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    auto pooled_off = [&](int row, int piece) {
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(max(row, 0)), p.d_k1));
        return static_cast<uint32_t>(pr * p.P2 + piece * (E / 2)) * static_cast<uint32_t>(sizeof(S));
    };
    auto pooled_rows = [&](int row) {
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(max(row, 0)), p.d_k1));
        return min(p.K1, O1 - pr * p.K1);   // (O1: the rows of the tensor the pool ran over -- the volume, or (CROP) the window)
    };
    const uint32_t pooled_plane_bytes = POOL ? static_cast<uint32_t>(p.P1) * static_cast<uint32_t>(p.P2) * static_cast<uint32_t>(sizeof(S)) : 0u;
    auto pooled_plane = [&](int pa, uint32_t &soff, int &n0) {   // unpooled plane (uniform) -> byte offset of its pooled plane, window planes
        const int pp = static_cast<int>(fdiv(static_cast<uint32_t>(max(pa, 0)), p.d_k0));
        soff = static_cast<uint32_t>(pp) * pooled_plane_bytes;
        n0 = min(p.K0, O0 - pp * p.K0);
    };
    // 8 pooled bytes -> the 16-byte piece of the unpooled gradient: every element twice, divided by the window size `cnt`
    auto expand = [&](u2 raw, int cnt) {
        Chunk<S, (E >= 2 ? E / 2 : 1)> in;
        __builtin_memcpy(in.e, &raw, 8);
        Chunk<S, E> out;
#pragma unroll
        for (int h = 0; h < E / 2; ++h) {
            const S q = narrow<T>(div_count<CT>(widen<T>(in.e[h]), cnt));
            out.e[2 * h] = q;
            out.e[2 * h + 1] = q;
        }
        u4 v;
        __builtin_memcpy(&v, out.e, 16);
        return v;
    };
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(gxp, 0, vol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, 0, kRsrcFlags);
    // 8 pooled bytes at byte `voff` (per thread; kOOR: none) + `soff` (uniform) of the (n, c) pooled volume
    auto pooled8 = [gmis](const __amdgpu_buffer_rsrc_t res, uint32_t voff, uint32_t soff) __attribute__((always_inline)) {
        if constexpr (ODD16) {
            const uint32_t eff = voff + (soff + gmis);           // (an out-of-range voff stays out of range: the volume is < 2^31 bytes)
            const u3 d = __builtin_amdgcn_raw_buffer_load_b96(res, eff & ~3u, 0u, 0);
            const uint32_t sh = (eff & 2u) << 3;                  // 0 or 16 bits
            return u2{__builtin_amdgcn_alignbit(d.y, d.x, sh), __builtin_amdgcn_alignbit(d.z, d.y, sh)};
        } else {
            return __builtin_amdgcn_raw_buffer_load_b64(res, voff, soff, 0);
        }
    };
    const uint32_t vx_own = sx_own >= 0 ? ox_own : kOOR;
    const uint32_t vg_own = sg_own >= 0 ? (POOL ? pooled_off(sg_own, tc) : og_own) : kOOR;
    const int n1_own = POOL ? pooled_rows(sg_own) : 1;   // window rows of the staged piece
    char *dump = tile + 2 * GP0 * 16 + tid * 16;
    char *dx_own = own ? tile + tid * 16 : dump;
    char *dg_own = own ? tile + (GP0 + tid) * 16 : dump;
    struct Staged {
        u4 xo, go;
        u2 po;    // POOL: the pooled bytes of the gradient piece, expanded when parked
        int n0;   // ... and the window planes of their pooled plane
    };
    auto load_planes = [&](int pax, int pag, Staged &v) {   // source planes (uniform; -1: fill)
        const uint32_t sx = pax >= 0 ? static_cast<uint32_t>(pax) * plane_bytes : 0u;
        v.xo = __builtin_amdgcn_raw_buffer_load_b128(pax >= 0 ? xres : none, vx_own, sx, 0);
        if constexpr (POOL) {
            uint32_t sg;
            pooled_plane(pag, sg, v.n0);
            v.po = pooled8(pag >= 0 ? gres : none, vg_own, sg);
        } else {
            const uint32_t sg = pag >= 0 ? static_cast<uint32_t>(pag) * gplane_bytes : 0u;
            v.go = __builtin_amdgcn_raw_buffer_load_b128(pag >= 0 ? gres : none, vg_own, sg, 0);
        }
    };
    auto park = [&](const Staged &v) {
        *reinterpret_cast<u4 *>(__builtin_assume_aligned(dx_own, 16)) = v.xo;
        if constexpr (POOL) *reinterpret_cast<u4 *>(__builtin_assume_aligned(dg_own, 16)) = expand(v.po, v.n0 * n1_own * 2);
        else *reinterpret_cast<u4 *>(__builtin_assume_aligned(dg_own, 16)) = v.go;
    };
    const bool mine = tr < R && tr < Rn;
    const int b = b0 + tr;
    const char *rows_x = tile + tr * RB, *rows_g = tile + (R + 1 + tr) * RB;
    const CT dw[3] = {static_cast<CT>(d.dw[0]), static_cast<CT>(d.dw[1]), static_cast<CT>(d.dw[2])};
    // ---- window reads: two aligned 16-byte spans per row and a uniform phase (lds_window6) ------------------------------
    // Fill rows and fill planes are zeros in the tile (empty resource / out-of-range offset), so only the columns are masked.
    // Zeros padding: every chunk is affine (one whose window lies outside the row has all its columns masked: any dwords do);
    // the other paddings: chunks whose map is not affine, or not at the workgroup's phase, read element by element.
    constexpr int ES = static_cast<int>(sizeof(S));
    if (!mine) {   // a thread without a chunk reads rows that are not its own: every column masked (0 * garbage is not 0)
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = gm.cm[e] = -1;
    }
    const int phx = (-d.cx2 * ES) & 15, phg = (-(d.cg2 + L2) * ES) & 15;
    const bool fx = PAD == 0 || (xm.affine && ((xm.base * ES) & 15) == phx);
    const bool fg = PAD == 0 || (gm.affine && ((gm.base * ES) & 15) == phg);
    uint32_t xmask[5] = {0, 0, 0, 0, 0}, gmask[5] = {0, 0, 0, 0, 0};   // 16-bit data: per-dword column masks
    if constexpr (ES == 2) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int hi = 2 * i + 1 <= E ? 2 * i + 1 : E;
            xmask[i] = (xm.cm[2 * i] >= 0 ? 0xffffu : 0u) | ((2 * i + 1 <= E && xm.cm[hi] >= 0) ? 0xffff0000u : 0u);
            gmask[i] = (gm.cm[2 * i] >= 0 ? 0xffffu : 0u) | ((2 * i + 1 <= E && gm.cm[hi] >= 0) ? 0xffff0000u : 0u);
        }
    }
    auto window_packed = [&](const char *rowp, const ColState<E> &cst, bool fast, int ph, const uint32_t(&m)[5], uint32_t(&t)[5]) {
        if (fast) {
            uint32_t o[6];
            lds_window6(rowp, cst.base * 2, ph, o);
            if (ph & 2) {   // uniform
#pragma unroll
                for (int i = 0; i < 5; ++i) t[i] = __builtin_amdgcn_alignbit(o[i + 1], o[i], 16) & m[i];
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) t[i] = o[i] & m[i];
            }
        } else if constexpr (PAD != 0) {
            const uint16_t *p0 = reinterpret_cast<const uint16_t *>(rowp);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int hi = 2 * i + 1 <= E ? 2 * i + 1 : E;
                const uint32_t lo = p0[cst.cm[2 * i] > 0 ? cst.cm[2 * i] : 0];
                const uint32_t up = p0[cst.cm[hi] > 0 ? cst.cm[hi] : 0];
                t[i] = (lo | (up << 16)) & m[i];
            }
        }
    };
    auto window = [&](const char *rowp, const ColState<E> &cst, bool fast, int ph, const uint32_t(&m)[5], S(&raw)[E + 1]) {
        if constexpr (ES == 2) {
            uint32_t t[5];
            window_packed(rowp, cst, fast, ph, m, t);
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                const uint16_t h = static_cast<uint16_t>(e & 1 ? t[e >> 1] >> 16 : t[e >> 1]);
                __builtin_memcpy(&raw[e], &h, 2);
            }
        } else {
            S zero;
            __builtin_memset(&zero, 0, sizeof(S));
            if (fast) {
                uint32_t o[6];
                lds_window6(rowp, cst.base * ES, ph, o);
#pragma unroll
                for (int e = 0; e <= E; ++e) {
                    if constexpr (ES == 4) {
                        __builtin_memcpy(&raw[e], &o[e], 4);
                    } else {
                        const uint64_t q = static_cast<uint64_t>(o[2 * e]) | (static_cast<uint64_t>(o[2 * e + 1]) << 32);
                        __builtin_memcpy(&raw[e], &q, 8);
                    }
                }
            } else {
                const S *p0 = reinterpret_cast<const S *>(rowp);
#pragma unroll
                for (int e = 0; e <= E; ++e) raw[e] = p0[cst.cm[e] > 0 ? cst.cm[e] : 0];
            }
#pragma unroll
            for (int e = 0; e <= E; ++e) raw[e] = cst.cm[e] >= 0 ? raw[e] : zero;
        }
    };

    // 16-bit data: the x corners never leave their packed form (the weight-gradient sums are v_dot2c products of packed pairs
    // of x and of the incoming gradient: 48 instructions per chunk instead of 128 subtractions and multiply-adds plus the
    // unpacking); cxm: per-dword masks of the window's columns
    constexpr bool PACKED = sizeof(S) == 2;
    constexpr int NS = PACKED ? 8 : NDIFF;   // running sums: per corner (packed) / per corner difference
    uint32_t cxp[2][5];
    CT cx[PACKED ? 1 : 2][PACKED ? 1 : E + 1], cg[2][E + 1];   // the "+0" planes' corner rows
    {
        const int pax0 = row_map_t<PAD>(a0, d.cx0, S0, p.pad), pag0 = ACTIVE ? row_map_t<PAD>(a0 - L0, d.cg0, O0, p.pad) : -1;
        Staged v0;
        load_planes(pax0, pag0, v0);
        park(v0);
        __syncthreads();
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if constexpr (ACTIVE) {
                S rg[E + 1];
                window(rows_g + hb * RB, gm, fg, phg, gmask, rg);
#pragma unroll
                for (int e = 0; e <= E; ++e) cg[hb][e] = widen<T>(rg[e]);
            }
            if constexpr (PACKED) {
                window_packed(rows_x + hb * RB, xm, fx, phx, xmask, cxp[hb]);
            } else {
                S rx[E + 1];
                window(rows_x + hb * RB, xm, fx, phx, xmask, rx);
#pragma unroll
                for (int e = 0; e <= E; ++e) cx[hb][e] = widen<T>(rx[e]);
            }
        }
    }
    // running sums: fp32 per step, folded every four planes into this thread's fp64 slots in LDS (registers are what limits
    // the number of resident workgroups here)
    double *accs = reinterpret_cast<double *>(tile + 2 * GP0 * 16 + kThreads * 16) + tid;   // [NS][kThreads]
    CT part[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        accs[i * kThreads] = 0.0;
        part[i] = CT(0);
    }
    const uint32_t my = mine ? static_cast<uint32_t>(b * S2 + ji) * static_cast<uint32_t>(sizeof(S)) : kOOR;   // own chunk, bytes within a plane
    const bool mine_g = mine && static_cast<unsigned>(b - L1) < static_cast<unsigned>(O1);   // CROP: the own row lies in the window
    const uint32_t myg = mine_g ? static_cast<uint32_t>((b - L1) * O2 + ji) * static_cast<uint32_t>(sizeof(S)) : kOOR;
    const uint32_t myg_prev = (mine_g && tc > 0 && L2 > 0) ? myg - 16u : kOOR;
    const uint32_t myp = (POOL && mine_g) ? pooled_off(b - L1, tc) : kOOR;   // POOL: its pooled bytes (CROP: of the piece at the own window columns)
    const uint32_t myp_prev = (POOL && CROP && mine_g && tc > 0 && L2 > 0) ? myp - 8u : kOOR;
    const int n1_my = POOL ? pooled_rows(max(b - L1, 0)) : 1;
    auto load_own = [&](int a, bool have) {   // the incoming gradient at the thread's own chunk of plane a (raw: u4, or the pooled 8 bytes in .xy)
        u4 r;
        if constexpr (POOL) {
            uint32_t sg;
            int n0;
            const bool in = have && (!CROP || static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0));   // (uniform)
            pooled_plane(a - L0, sg, n0);
            const u2 q = pooled8(in ? gres : none, myp, in ? sg : 0u);
            r = u4{q.x, q.y, static_cast<uint32_t>(n0), 0u};
        } else if constexpr (CROP) {   // (the piece at the thread's own window columns; its predecessor: load_own_prev)
            const bool in = have && static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0);   // (uniform)
            r = __builtin_amdgcn_raw_buffer_load_b128(in ? gres : none, myg, in ? static_cast<uint32_t>(a - L0) * gplane_bytes : 0u, 0);
        } else {
            r = __builtin_amdgcn_raw_buffer_load_b128(have ? gres : none, my, static_cast<uint32_t>(a) * plane_bytes, 0);
        }
        return r;
    };
    auto load_own_prev = [&](int a, bool have) {
        u4 r = u4{0u, 0u, 0u, 0u};
        if constexpr (CROP && POOL) {
            uint32_t sg;
            int n0;
            const bool in = have && static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0);
            pooled_plane(a - L0, sg, n0);
            const u2 q = pooled8(in ? gres : none, myp_prev, in ? sg : 0u);
            r = u4{q.x, q.y, static_cast<uint32_t>(n0), 0u};
        } else if constexpr (CROP) {
            const bool in = have && static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0);
            r = __builtin_amdgcn_raw_buffer_load_b128(in ? gres : none, myg_prev, in ? static_cast<uint32_t>(a - L0) * gplane_bytes : 0u, 0);
        }
        return r;
    };
    // CROP: the own chunk = window columns ji - L2 ..: L2 elements of the previous aligned piece in front (a funnel by whole dwords:
    // 4- / 8-byte elements), columns beyond the window's row and rows outside the window zero
    bool keep_col[E];
#pragma unroll
    for (int e = 0; e < E; ++e) keep_col[e] = !CROP || (mine_g && static_cast<unsigned>(ji + e - L2) < static_cast<unsigned>(O2));
    auto own_chunk = [&](const u4 &lo, const u4 &hi) {
        u4 r = hi;
        if constexpr (CROP) {
            const int sb = L2 * ES;   // bytes (uniform): 0, 2, 4, 8 or 16
            if (sb == 2) r = u4{__builtin_amdgcn_alignbit(hi.x, lo.w, 16), __builtin_amdgcn_alignbit(hi.y, hi.x, 16), __builtin_amdgcn_alignbit(hi.z, hi.y, 16),
                                __builtin_amdgcn_alignbit(hi.w, hi.z, 16)};
            else if (sb == 4) r = u4{lo.w, hi.x, hi.y, hi.z};
            else if (sb == 8) r = u4{lo.z, lo.w, hi.x, hi.y};
            else if (sb == 16) r = lo;
            Chunk<S, E> c;
            __builtin_memcpy(c.e, &r, 16);
            S zero;
            __builtin_memset(&zero, 0, sizeof(S));
#pragma unroll
            for (int e = 0; e < E; ++e) c.e[e] = keep_col[e] ? c.e[e] : zero;
            __builtin_memcpy(&r, c.e, 16);
        }
        return r;
    };
    __syncthreads();   // the "+0" planes have been read
    constexpr int GA = ACTIVE ? 1 : 0;   // the gradient plane of step a: the "+1" corner plane / the plane the tap reads
    // the planes of steps a0 and a0 + 1 (steps that do not exist: empty resources; a buffer's range check does not see the
    // scalar offset)
    Staged stA, stB;
    load_planes(row_map_t<PAD>(a0 + 1, d.cx0, S0, p.pad), row_map_t<PAD>(a0 + GA - L0, d.cg0, O0, p.pad), stA);
    load_planes(a0 + 1 < a1 ? row_map_t<PAD>(a0 + 2, d.cx0, S0, p.pad) : -1, a0 + 1 < a1 ? row_map_t<PAD>(a0 + 1 + GA - L0, d.cg0, O0, p.pad) : -1, stB);
    u4 gcur = load_own(a0, true), gprev = load_own_prev(a0, true);
    auto walk_step = [&](int a, Staged &pend) {   // `pend` holds the planes of step a; it leaves with those of step a + 2 in flight
        park(pend);
        __syncthreads();
        const bool more = a + 2 < a1;
        load_planes(more ? row_map_t<PAD>(a + 3, d.cx0, S0, p.pad) : -1, more ? row_map_t<PAD>(a + 2 + GA - L0, d.cg0, O0, p.pad) : -1, pend);
        Chunk<S, E> gch;
        if constexpr (POOL && CROP) {
            const u4 oc = own_chunk(expand(u2{gprev.x, gprev.y}, static_cast<int>(gprev.z) * n1_my * 2), expand(u2{gcur.x, gcur.y}, static_cast<int>(gcur.z) * n1_my * 2));
            __builtin_memcpy(gch.e, &oc, 16);
        } else if constexpr (POOL) {
            const u4 ex = expand(u2{gcur.x, gcur.y}, static_cast<int>(gcur.z) * n1_my * 2);
            __builtin_memcpy(gch.e, &ex, 16);
        } else if constexpr (CROP) {
            const u4 oc = own_chunk(gprev, gcur);
            __builtin_memcpy(gch.e, &oc, 16);
        } else {
            __builtin_memcpy(gch.e, &gcur, 16);
        }
        // ---- weight-gradient sums: corners of x (plane bit 0, row bit 1, column bit 2: step_backward's order) ------------
        if constexpr (PACKED) {
            uint32_t gq[4], nxp[2][5];
            __builtin_memcpy(gq, gch.e, 16);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                window_packed(rows_x + hb * RB, xm, fx, phx, xmask, nxp[hb]);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const uint32_t(&wv)[5] = pl ? nxp[hb] : cxp[hb];
                    const int q0 = pl | (hb << 1), q1 = q0 | 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        part[q0] = dot2_packed<T>(gq[i], wv[i], part[q0]);
                        part[q1] = dot2_packed<T>(gq[i], __builtin_amdgcn_alignbit(wv[i + 1], wv[i], 16), part[q1]);
                    }
                }
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int i = 0; i < 5; ++i) cxp[hb][i] = nxp[hb][i];
        } else {
            CT nx[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                S rx[E + 1];
                window(rows_x + hb * RB, xm, fx, phx, xmask, rx);
#pragma unroll
                for (int e = 0; e <= E; ++e) nx[hb][e] = widen<T>(rx[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[8], df[NDIFF];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int hb = (q >> 1) & 1, col = e + (q >> 2);
                    v[q] = (q & 1) ? nx[hb][col] : cx[hb][col];
                }
                corner_diffs<3, CT>(v, df);
                const CT gval = widen<T>(gch.e[e]);
#pragma unroll
                for (int i = 0; i < NDIFF; ++i) part[i] = fma_ct(gval, df[i], part[i]);
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int e = 0; e <= E; ++e) cx[hb][e] = nx[hb][e];
        }
        // the next step's own chunk: in flight through the blends below and the next step's staging
        gcur = load_own(a + 1, a + 1 < a1);
        if constexpr (CROP) gprev = load_own_prev(a + 1, a + 1 < a1);
        // ---- grad_x ------------------------------------------------------------------------------------------------
        Chunk<S, E> res;
        if constexpr (ACTIVE) {
            CT rowb[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                S rg[E + 1];
                window(rows_g + hb * RB, gm, fg, phg, gmask, rg);
#pragma unroll
                for (int e = 0; e <= E; ++e) {
                    const CT nv = widen<T>(rg[e]);
                    const CT two[2] = {cg[hb][e], nv};
                    rowb[hb][e] = interp_t<T, 1>(two, &dw[0]);
                    cg[hb][e] = nv;
                }
            }
            CT colb[E + 1];
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                const CT two[2] = {rowb[0][e], rowb[1][e]};
                colb[e] = interp_t<T, 1>(two, &dw[1]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const CT two[2] = {colb[e], colb[e + 1]};
                res.e[e] = narrow<T>(interp_t<T, 1>(two, &dw[2]));
            }
        } else if constexpr (PACKED) {   // the sparse shift: a raw copy of the window (the bit pattern is kept)
            uint32_t t[5];
            window_packed(rows_g, gm, fg, phg, gmask, t);
            __builtin_memcpy(res.e, t, 16);
        } else {
            S rg[E + 1];
            window(rows_g, gm, fg, phg, gmask, rg);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = rg[e];
        }
        if constexpr (CROP) {   // grad_x is zero outside the window
            const bool in_plane = static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0);   // (uniform)
            S zero;
            __builtin_memset(&zero, 0, sizeof(S));
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = (in_plane && keep_col[e]) ? res.e[e] : zero;
        }
        {
            u4 bits;
            __builtin_memcpy(&bits, res.e, 16);
            buffer_store_b128_soffset<0>(bits, ores, my, static_cast<uint32_t>(a) * plane_bytes);
        }
        if ((a & 3) == 3 || a == a1 - 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                accs[i * kThreads] += static_cast<double>(part[i]);
                part[i] = CT(0);
            }
        }
        __syncthreads();   // everybody has read this step's planes
    };
    int a = a0;
    for (; a + 1 < a1; a += 2) {   // whole pairs: no condition between the steps (exact wait counts)
        walk_step(a, stA);
        walk_step(a + 1, stB);
    }
    if (a < a1) walk_step(a, stA);
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = accs[i * kThreads];
    if constexpr (PACKED) {  // per-corner sums -> the corner-difference sums (corner_diffs is linear)
        double v[8], df[NDIFF];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = acc[q];
        corner_diffs<3, double>(v, df);
#pragma unroll
        for (int i = 0; i < NDIFF; ++i) acc[i] = df[i];
    }
    // ---- the workgroup's sums: shuffle tree per wave, the four waves added by one thread ----------------------------------
    __syncthreads();
    double *scratch = reinterpret_cast<double *>(tile);   // the tile is dead
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) {
        const double t = wave_total(acc[i]);
        if ((tid & 63) == 63) scratch[NDIFF * wave + i] = t;
    }
    __syncthreads();
    if (tid < NDIFF) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) sum += scratch[NDIFF * w + tid];
        p.partials[static_cast<size_t>(bid) * NDIFF + tid] = sum;
    }
}

}  // namespace

// the 3-D interpolating forward as a walk through the planes: contiguous, no crop, rows of whole 16-byte pieces and at most
// one workgroup pass wide
static bool walk_forward_core(const Geometry &g, int dtype, const void *x, const void *out, bool pooled);
bool walk_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    return g.K[0] <= 0 && walk_forward_core(g, dtype, x, out, false);
}
// the fused shift + average pool forward in 3-D (interpolating): windows (K0, K1 <= 2, 2), `out` = the pooled tensor, contiguous
bool walk_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (!(g.K[0] > 0 && g.nd == 3) || g.K[2] != 2 || g.K[1] < 1 || g.K[1] > 2) return false;
    return walk_forward_core(g, dtype, x, out, true);
}
static bool walk_forward_core(const Geometry &g, int dtype, const void *x, const void *out, bool pooled) {
    if (g_step_tune[2] == 1 || (g_step_tune[3] & 16)) return false;   // knob 34 = 1: no forwards through LDS; knob 35 bit 4: no walk
    if (dtype > SHIFTND_BF16 || g.nd != 3 || !g.active) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2];
    if (xe < 1 || xe >= (1LL << 30) || g.S[2] > 32000 || g.S[0] < 2) return false;
    if ((g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || (!pooled && !dense(g.os, g.N, g.C, g.O))) return false;
    const int64_t cpr = g.S[2] * es / 16;
    if (cpr > kThreads / 2 || g.S[0] * g.S[1] * g.S[2] * es >= (1LL << 31)) return false;   // (one piece per thread; one buffer resource per volume)
    const int64_t rmax = std::min<int64_t>(kThreads / cpr - 1, g.S[1]);
    if (64 + (rmax + 2) * cpr * 16 + 64 + 2 * kThreads * 16 + 64 + kThreads * 16 > 64 * 1024) return false;
    const int64_t spp = (g.S[1] + rmax - 1) / rmax;
    if (g.N * g.C * (spp + 1) + 8 >= (1LL << 31)) return false;
    if (pooled) return rmax >= 2 || g.K[1] == 1;   // (a window's two rows live in one workgroup)
    // same box, N8 C128 16x112x112: fp32 0.308 (step_forward_lds) -> 0.259 ms; fp64 on request (knob 35 bit 5); 16-bit tensors only
    // with a fused pool (the plain calls: walk_forward16, shiftnd_walk.hip)
    if (es == 2) return pooled;
    return es == 4 || (g_step_tune[3] & 32);
}

int walk_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    FwdParams p{};
    p.pad = g.pad;
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.S0 = p.O0 = static_cast<int>(g.S[0]);
    p.S1 = p.O1 = static_cast<int>(g.S[1]);
    p.S2 = p.O2 = static_cast<int>(g.S[2]);
    p.x_plane = p.o_plane = g.S[0] * g.S[1] * g.S[2];
    p.cpr = p.xppr = static_cast<int>(g.S[2] * es / 16);
    const int rmax = std::min<int>(kThreads / p.cpr - 1, p.S1);   // (R + 1) * cpr <= 256: every staged piece has its thread
    p.spp = (p.S1 + rmax - 1) / rmax;
    p.R = (p.S1 + p.spp - 1) / p.spp;   // balanced steps: 112 rows of 14 pieces -> 7 steps of 16 rows, not 6 of 18 and one of 4
    const bool pooled = g.K[0] > 0;
    if (pooled) {
        p.K0 = static_cast<int>(g.K[0]);
        p.K1 = static_cast<int>(g.K[1]);
        p.P1 = static_cast<int>(g.P[1]);
        p.P2 = static_cast<int>(g.P[2]);
        p.p_plane = g.P[0] * g.P[1] * g.P[2];
        p.d_k0 = make_fastdiv(static_cast<uint32_t>(p.K0));
        if (p.K1 == 2 && (p.R & 1)) {   // a window's two rows in one workgroup: an even number of rows per step
            p.R = p.R + 1 <= rmax ? p.R + 1 : p.R - 1;
            p.spp = (p.S1 + p.R - 1) / p.R;
        }
    }
    p.spv = p.spp;
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_spv = p.d_spp;
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(p.cpr));
    p.d_xppr = p.d_cpr;
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const size_t lds = 64 + static_cast<size_t>(p.R + 1) * p.cpr * 16 + 64 + (pooled ? 2 * kThreads * 16 + 64 : 0) + kThreads * 16;   // tile (+ exchange slots), dump slots
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    note_kernel(pooled ? "walk_forward_pool" : "walk_forward");
#define SHIFTND_WALK_FWD_PAD(T, PADV) \
    case PADV: \
        if (pooled) hipLaunchKernelGGL((walk_forward<T, PADV, true>), grid, block, lds, st, p); \
        else if constexpr (sizeof(typename T::S) != 2) hipLaunchKernelGGL((walk_forward<T, PADV, false>), grid, block, lds, st, p); \
        break;
#define SHIFTND_WALK_FWD(T) \
    switch (g.pad) { SHIFTND_WALK_FWD_PAD(T, 0) SHIFTND_WALK_FWD_PAD(T, 1) SHIFTND_WALK_FWD_PAD(T, 2) default: SHIFTND_WALK_FWD_PAD(T, 3) }
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_WALK_FWD(f32_t) break;
    case SHIFTND_F64: SHIFTND_WALK_FWD(f64_t) break;
    case SHIFTND_F16: SHIFTND_WALK_FWD(f16_t) break;
    default: SHIFTND_WALK_FWD(bf16_t) break;
    }
#undef SHIFTND_WALK_FWD
#undef SHIFTND_WALK_FWD_PAD
    return SHIFTND_OK;
}

// the 3-D interpolating backward as a walk through the planes (walk_backward): what step_backward takes, 3-D, >= 2 planes.
// Automatic for 2- and 4-byte elements; knob 35 bit 5 (32): fp64 too; bit 4 (16): never; bit 0: the one-step form instead.
static bool walk_backward_core(const Geometry &g, int dtype, const void *go, const void *x, const void *gx, bool pooled);
bool walk_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    return g.K[0] <= 0 && walk_backward_core(g, dtype, go, x, gx, false);
}
// the fused shift + average-pool backward in 3-D (both shifts): `go` = gradient of the pooled output, contiguous; windows
// (K0, K1, 2)
bool walk_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (!(g.K[0] > 0 && g.nd == 3) || g.K[2] != 2 || g.K[1] < 1) return false;
    if (reinterpret_cast<uintptr_t>(go) % 8) return false;
    if (g.P[0] * g.P[1] * g.P[2] * dtype_size(dtype) >= (1LL << 31)) return false;
    return walk_backward_core(g, dtype, nullptr, x, gx, true);
}
static bool walk_backward_core(const Geometry &g, int dtype, const void *go, const void *x, const void *gx, bool pooled) {
    if (g_step_tune[0] == 1 || (g_step_tune[3] & 16)) return false;   // knob 32 = 1: never; knob 35 bit 4: no walk kernels
    if (dtype > SHIFTND_BF16 || g.nd != 3 || g.S[0] < 2) return false;
    const int es = dtype_size(dtype);
    bool cropped = false;
    for (int d = 0; d < 3; ++d) cropped = cropped || g.O[d] != g.S[d] || g.L[d] != 0;
    // (a window: walk_backward<.., CROP> -- zeros padding; 4-byte elements, or -- with the pool riding on the walk -- 2-byte ones too:
    //  the pooled gradient of a window with rows of an even number of elements)
    if (cropped && (es == 8 || (!pooled && es != 4) || !walk_crop_window_ok(g, pooled) || (pooled && g.O[2] % 2 != 0))) return false;
    if (g.S[1] < 1 || (g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads || g.S[2] > 32000) return false;
    if (g.S[0] * g.S[1] * g.S[2] >= (1LL << 30)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || (!pooled && !dense(g.os, g.N, g.C, g.O)) || !dense(g.gs, g.N, g.C, g.S)) return false;
    if ((!pooled && reinterpret_cast<uintptr_t>(go) % 16) || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(gx) % 16) return false;
    const StepLayout L = step_layout(g, es);
    if (L.total_steps + 8 >= (1ull << 31)) return false;
    if (L.cpr > kThreads / 2) return false;   // (R + 1 rows of pieces per plane and tensor, one piece per thread)
    const int64_t rmax = std::min<int64_t>(kThreads / L.cpr - 1, g.S[1]);
    if (64 + 2 * (rmax + 1) * L.cpr * 16 + kThreads * 16 + kThreads * 64 + 64 > 64 * 1024) return false;
    if (g.S[0] * g.S[1] * g.S[2] * es >= (1LL << 31)) return false;   // (one buffer resource spans an (n, c) volume)
    // same box, N8 C128 16x112x112: fp32 0.524 (slide_backward) -> 0.486 ms; fp64 on request (bit 5); 16-bit tensors only with a
    // fused pool (the plain calls: walk_backward16, shiftnd_walk.hip)
    if (es == 2) return pooled;
    return es == 4 || (g_step_tune[3] & 32);
}

template <typename T> static void launch_walk_backward(StepParams &p, size_t lds, bool active, void *gw, hipStream_t st) {
    using S = typename T::S;
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    launch_step_prep(T::kDtype, active, p, st);
    // (16-bit tensors without a pool: walk_backward16, shiftnd_walk.hip -- only their pooled variants are instantiated here)
    constexpr bool PLAIN = sizeof(S) != 2;
#define SHIFTND_WALK_BWD(PADV) \
    case PADV: \
        if (p.crop && p.K0 > 0) { \
            if constexpr (PADV == 0 && sizeof(S) != 8) { \
                if (!active) hipLaunchKernelGGL((walk_backward<T, 0, true, false, true>), grid, block, lds, st, p); \
                else hipLaunchKernelGGL((walk_backward<T, 0, true, true, true>), grid, block, lds, st, p); \
            } \
        } else if (p.crop) { \
            if constexpr (PADV == 0 && sizeof(S) == 4) { \
                if (!active) hipLaunchKernelGGL((walk_backward<T, 0, false, false, true>), grid, block, lds, st, p); \
                else hipLaunchKernelGGL((walk_backward<T, 0, false, true, true>), grid, block, lds, st, p); \
            } \
        } else if (!active && p.K0 > 0) hipLaunchKernelGGL((walk_backward<T, PADV, true, false>), grid, block, lds, st, p); \
        else if (p.K0 > 0) hipLaunchKernelGGL((walk_backward<T, PADV, true>), grid, block, lds, st, p); \
        else if constexpr (PLAIN) { \
            if (!active) hipLaunchKernelGGL((walk_backward<T, PADV, false, false>), grid, block, lds, st, p); \
            else hipLaunchKernelGGL((walk_backward<T, PADV, false>), grid, block, lds, st, p); \
        } \
        break;
    switch (p.pad) { SHIFTND_WALK_BWD(0) SHIFTND_WALK_BWD(1) SHIFTND_WALK_BWD(2) default: SHIFTND_WALK_BWD(3) }
#undef SHIFTND_WALK_BWD
    launch_step_reduce(T::kDtype, 3, p, gw, st);
}

// the second half of step_backward()'s launch for 3-D problems: balanced row steps, one record of sums per workgroup
int walk3_backward_launch(StepParams &p, const Geometry &g, int dtype, int cpr, void *gw, hipStream_t st) {
    const int es = dtype_size(dtype);
    struct { int cpr; } L{cpr};

        // the walk through the planes: balanced row steps, one record of sums per workgroup
        if (g.K[0] > 0) {
            p.K0 = static_cast<int>(g.K[0]);
            p.P0 = static_cast<int>(g.P[0]);
            p.d_k0 = make_fastdiv(static_cast<uint32_t>(p.K0));
            p.g_plane = g.P[0] * g.P[1] * g.P[2];
        }
        bool crop = false;
        for (int d = 0; d < 3; ++d) crop = crop || g.O[d] != g.S[d] || g.L[d] != 0;
        if (crop) {   // walk_backward<.., CROP>: the window's sizes in wO0 / wO1 / wO2, its first plane / row / column in wL0 / wL1 / wL2
            p.crop = 1;
            p.wO0 = static_cast<int>(g.O[0]);
            p.wO1 = static_cast<int>(g.O[1]);
            p.wO2 = static_cast<int>(g.O[2]);
            p.wL0 = static_cast<int>(g.L[0]);
            p.wL1 = static_cast<int>(g.L[1]);
            p.wL2 = static_cast<int>(g.L[2]);
            if (g.K[0] <= 0) p.g_plane = g.O[0] * g.O[1] * g.O[2];   // (pooled: the pooled window's elements, set above)
        }
        const int rmax = std::min<int>(kThreads / L.cpr - 1, p.S1);   // (R + 1) * cpr <= 256: every staged piece has its thread
        p.spp = (p.S1 + rmax - 1) / rmax;
        p.R = (p.S1 + p.spp - 1) / p.spp;
        // planes per workgroup (knob 38): all of them, or a part of the depth -- more, shorter workgroups, one more staged plane each
        p.walk_planes = (g_step_tune[4] > 0 && g_step_tune[4] < p.S0) ? g_step_tune[4] : p.S0;
        const int dparts = (p.S0 + p.walk_planes - 1) / p.walk_planes;
        p.spv = dparts * p.spp;
        uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spv;
        if (total > step_layout(g, es, 1).total_steps) {
            // more records than the workspace was planned for (knob 38 with few rows per step: a walk step holds one row fewer
            // than a one-step workgroup): walk the whole depth -- one part, at most twice the one-step plan's row steps <= S0 of them
            p.walk_planes = p.S0;
            p.spv = p.spp;
            total = static_cast<uint64_t>(g.N) * g.C * p.spv;
        }
        p.total_steps = static_cast<uint32_t>(total);
        p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
        p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
        p.d_spv = make_fastdiv(static_cast<uint32_t>(p.spv));
        const size_t lds = 64 + static_cast<size_t>(2 * (p.R + 1)) * L.cpr * 16 + kThreads * 16 + kThreads * 8 * sizeof(double) + 64;   // tile, dump slots, sums, pad
        note_kernel(g.K[0] > 0 ? (crop ? "walk_backward_crop_pool" : "walk_backward_pool") : (crop ? (g.active ? "walk_backward_crop" : "walk_backward_crop_sparse") : (g.active ? "walk_backward" : "walk_backward_sparse")));
        switch (dtype) {
        case SHIFTND_F32: launch_walk_backward<f32_t>(p, lds, g.active != 0, gw, st); break;
        case SHIFTND_F64: launch_walk_backward<f64_t>(p, lds, g.active != 0, gw, st); break;
        case SHIFTND_F16: launch_walk_backward<f16_t>(p, lds, g.active != 0, gw, st); break;
        default: launch_walk_backward<bf16_t>(p, lds, g.active != 0, gw, st); break;
        }
        return SHIFTND_OK;
    }

}  // namespace shiftnd
