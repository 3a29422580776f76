// shiftnd_step_fwd.hip -- the forwards of the one-step family (split from shiftnd_step.hip in round 4 so that the build
// parallelises): step_gather_forward (sparse-shift / quantized forward of 4- / 8-byte elements, 2-D and
// 3-D, no LDS), step_gather_forward_small (1- / 2-byte elements: two aligned loads and a uniform byte funnel),
// step_gather_forward_pool (the module's 2 x 2 average pool as the epilogue), step_forward_lds (interpolating forward of every
// float dtype, sparse shift of 2-byte elements: rows through LDS).
// Reference behaviour restated: kernels/shifts_kernels.h:156-220, :532-571; cuda/shifts_cuda.cu:168-183.
#include "shiftnd_step.hpp"

namespace shiftnd {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// step_gather_forward: the sparse-shift / quantized forward of 4- and 8-byte elements as the same linear sweep of
// one-step workgroups (sweep_gather_forward with one row group per workgroup reaches 3.9 TB/s: its generic per-wave
// prologue is what a short workgroup cannot afford).  No LDS, no barrier, no table: a thread loads its 16-byte chunk
// at the shifted position (gfx950 global loads take any alignment) and stores it; the padding mode is a template
// parameter, the channel's shifts come from one scalar load of its weights.
// ---------------------------------------------------------------------------------------------------------------------
// ND = 3 (float weights): the step is (n, c, output plane a, row step); the source plane of a is one more folded index.
template <int ESIZE, int PAD, int ND = 2>
__global__ __launch_bounds__(kThreads) void step_gather_forward(const GatherParams p) {
    using R_t = typename raw_t<ESIZE>::type;
    constexpr int E = 16 / ESIZE;
    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    uint32_t plane;
    int step, a = 0, pa = 0, cs1, cs2;
    if constexpr (ND == 3) {
        plane = fdiv(bid, p.d_spv);
        const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spv);
        a = static_cast<int>(fdiv(vstep, p.d_spp));
        step = static_cast<int>(vstep) - a * p.spp;
        const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
        int cs0;
        if (p.wkind == SHIFTND_F64) {
            double wv[3];
            load_weights_nd<double>(p.w, p.wkind, c, 3, wv);
            cs0 = canon_of<PAD, double>(rint(wv[0]), p.S0, p.d_per0, p.pad);
            cs1 = canon_of<PAD, double>(rint(wv[1]), p.S1, p.d_per1, p.pad);
            cs2 = canon_of<PAD, double>(rint(wv[2]), p.S2, p.d_per2, p.pad);
        } else {
            float wv[3];
            load_weights_nd<float>(p.w, p.wkind, c, 3, wv);
            cs0 = canon_of<PAD, float>(rintf(wv[0]), p.S0, p.d_per0, p.pad);
            cs1 = canon_of<PAD, float>(rintf(wv[1]), p.S1, p.d_per1, p.pad);
            cs2 = canon_of<PAD, float>(rintf(wv[2]), p.S2, p.d_per2, p.pad);
        }
        cs0 = __builtin_amdgcn_readfirstlane(cs0);
        cs1 = __builtin_amdgcn_readfirstlane(cs1);
        cs2 = __builtin_amdgcn_readfirstlane(cs2);
        pa = row_map_t<PAD>(a + p.L0, cs0, p.S0, p.pad);
    } else {
        plane = fdiv(bid, p.d_spp);
        step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
        const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
        channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2, p.pad);
    }
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int r = step * p.R + tr;
    if (tr >= p.R || r >= p.O1) return;
    const int jo = tc * E;
    const int rb = row_map_t<PAD>(r + p.L1, cs1, p.S1, p.pad);
    int mm[E];
    bool contig = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        mm[e] = row_map_t<PAD>(jo + p.L2 + e, cs2, p.S2, p.pad);
        contig = contig && (mm[e] == mm[0] + e);
    }
    contig = contig && mm[0] >= 0;
    const R_t fill = static_cast<R_t>(p.fill);
    const R_t *xp = static_cast<const R_t *>(p.x) + static_cast<int64_t>(plane) * p.x_plane + static_cast<int64_t>(pa < 0 ? 0 : pa) * p.S1 * p.S2;
    R_t *dst = static_cast<R_t *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + (static_cast<int64_t>(a) * p.O1 + r) * p.O2 + jo;
    Chunk<R_t, E> v;
    if (rb < 0 || pa < 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) v.e[e] = fill;
    } else {
        const R_t *row = xp + rb * p.S2;
        if (contig) {
            v = load_chunk<R_t, E, true>(row + mm[0]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) v.e[e] = mm[e] >= 0 ? __builtin_nontemporal_load(row + mm[e]) : fill;
        }
    }
    store_chunk<R_t, E>(dst, v);
}


// The same for 1- and 2-byte elements, where a 16-byte load at element alignment is slow: the output chunk's 16 source
// bytes lie in two ALIGNED 16-byte pieces of the source row (rows are whole pieces), displaced by a byte phase that is
// the same for the whole workgroup (one channel = one inner shift; the crop offset is uniform): two aligned loads, a
// uniform switch on the dword part of the phase and one v_alignbit per output dword.  With zeros padding a piece is
// either inside the row or entirely fill, so the row ends need no element path at all; the wrapping / clamping paddings
// send only the chunks that touch a row end through the element-by-element map.
// step_gather_forward with the module's 2 x 2 average pool as its epilogue (2-D sparse shift, 4- / 8-byte float elements): a thread
// gathers the chunk of BOTH rows of a pooled row, sums each window in ATen's order (row, then column) in the compute type, divides
// by the window size and stores E / 2 pooled elements -- the shift output never exists.  `out` is the pooled tensor [N, C, P1, P2];
// p.O1 / p.O2 are the sizes of the (virtual) shift output, p.spp counts steps of R POOLED rows.
// the 16 bytes at byte phase `ph` (uniform) of the 32 bytes A | B (step_gather_forward_small's funnel)
typedef uint32_t gather_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gather_u4 funnel_bytes(const gather_u4 A, const gather_u4 B, const int ph) {
    const uint32_t sh = static_cast<uint32_t>(ph & 3) * 8u;
    switch (ph >> 2) {  // uniform
    case 0: return gather_u4{__builtin_amdgcn_alignbit(A.y, A.x, sh), __builtin_amdgcn_alignbit(A.z, A.y, sh), __builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh)};
    case 1: return gather_u4{__builtin_amdgcn_alignbit(A.z, A.y, sh), __builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh)};
    case 2: return gather_u4{__builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh), __builtin_amdgcn_alignbit(B.z, B.y, sh)};
    default: return gather_u4{__builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh), __builtin_amdgcn_alignbit(B.z, B.y, sh), __builtin_amdgcn_alignbit(B.w, B.z, sh)};
    }
}

// ACTIVE (round 6): the interpolating shift -- three source rows (the two of the pooled row and the + 1 corner row) of E + 1 columns
// per thread, the blends of interp_t<T, 2>, the shift's result rounded to the storage type like the two-step sequence, then the same
// sums (the band-walk kernel ran it at 2 - 2.9 TB/s).
template <typename T, int PAD, bool ACTIVE = false>
__global__ __launch_bounds__(kThreads) void step_gather_forward_pool(const GatherParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int NR = ACTIVE ? 3 : 2, NC = ACTIVE ? E + 1 : E;   // source rows / columns a thread reads
    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs1, cs2;
    CT dw[2] = {CT(0), CT(0)};
    if constexpr (ACTIVE) {
        CT wr, wc;
        load_weights2<CT>(p.w, p.wkind, c, wr, wc);
        const CT rr = c_floor<CT>(wr), rc = c_floor<CT>(wc);
        dw[0] = wr - rr;
        dw[1] = wc - rc;
        cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr, p.S1, p.d_per1, p.pad));
        cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rc, p.S2, p.d_per2, p.pad));
    } else {
        channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2, p.pad);
    }
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int P1 = (p.O1 + 1) >> 1, P2 = (p.O2 + 1) >> 1;   // (round 6: any window width -- a cropped 224 -> 222 is not whole pieces)
    const int pr = step * p.R + tr;   // pooled row
    if (tr >= p.R || pr >= P1) return;
    const int jo = tc * E;
    int mm[NC];
    bool contig = true;
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        mm[e] = row_map_t<PAD>(jo + p.L2 + e, cs2, p.S2, p.pad);
        contig = contig && (e >= E || mm[e] == mm[0] + e);   // (the first E columns: one 16-byte load; the corner column on its own)
    }
    contig = contig && mm[0] >= 0;
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const int n1 = min(2, p.O1 - 2 * pr);   // rows of this window row (a ragged last one: 1)
    S zero;
    __builtin_memset(&zero, 0, sizeof(S));
    S v[NR][NC];
    // 16-bit elements, sparse shift, zeros padding, source rows of whole aligned pieces (host: p.xppr > 0): an element-aligned 16-byte
    // load of 2-byte elements is slow (the 2.2 - 2.5 TB/s of this kernel in bf16) -- the chunk's 16 source bytes come from the two
    // ALIGNED pieces that hold them, displaced by a byte phase that is uniform for the workgroup (step_gather_forward_small's form);
    // a piece is either inside the row or entirely fill
    bool done16 = false;
    if constexpr (sizeof(S) == 2 && !ACTIVE && PAD == 0) {
        if (p.xppr > 0) {   // (uniform)
            const int dcol = p.L2 - cs2, ph = (dcol * 2) & 15, q = tc + ((dcol * 2) >> 4);
            const bool qa_in = q >= 0 && q < p.xppr, qb_in = q + 1 >= 0 && q + 1 < p.xppr;
            const int qa = q < 0 ? 0 : (q >= p.xppr ? p.xppr - 1 : q), qb = q + 1 < 0 ? 0 : (q + 1 >= p.xppr ? p.xppr - 1 : q + 1);
            const gather_u4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rb = h < n1 ? row_map_t<PAD>(2 * pr + h + p.L1, cs1, p.S1, p.pad) : -1;
                const gather_u4 *xrow = reinterpret_cast<const gather_u4 *>(xp + static_cast<int64_t>(rb < 0 ? 0 : rb) * p.S2);
                gather_u4 A = __builtin_nontemporal_load(xrow + qa), B = zero4;
                if (ph != 0) B = __builtin_nontemporal_load(xrow + qb);   // (uniform)
                A = (rb >= 0 && qa_in) ? A : zero4;
                B = (rb >= 0 && qb_in) ? B : zero4;
                const gather_u4 o = funnel_bytes(A, B, ph);
                __builtin_memcpy(v[h], &o, 16);
            }
            done16 = true;
        }
    }
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        if (done16) break;
        const int rb = h < n1 + (ACTIVE ? 1 : 0) ? row_map_t<PAD>(2 * pr + h + p.L1, cs1, p.S1, p.pad) : -1;
        if (rb < 0) {
#pragma unroll
            for (int e = 0; e < NC; ++e) v[h][e] = zero;
        } else {
            const S *row = xp + rb * p.S2;
            if (contig) {
                const Chunk<S, E> ch = load_chunk<S, E, true>(row + mm[0]);
#pragma unroll
                for (int e = 0; e < E; ++e) v[h][e] = ch.e[e];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) v[h][e] = mm[e] >= 0 ? __builtin_nontemporal_load(row + mm[e]) : zero;
            }
            if constexpr (ACTIVE) v[h][E] = mm[E] >= 0 ? row[mm[E]] : zero;
        }
    }
    // the shift's output at the window's positions (interpolating: rounded to the storage type, as the unfused sequence stores it)
    CT y[2][E];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (ACTIVE) {
                const CT q[4] = {widen<T>(v[h][e]), widen<T>(v[h + 1][e]), widen<T>(v[h][e + 1]), widen<T>(v[h + 1][e + 1])};
                y[h][e] = widen<T>(narrow<T>(interp_t<T, 2>(q, dw)));
            } else {
                y[h][e] = widen<T>(v[h][e]);
            }
        }
    Chunk<S, (E / 2 > 0 ? E / 2 : 1)> outc;
    // (sums in ATen's order: row, then column; a ragged last window -- an odd window width or height -- has one column / row)
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const bool two = jo + 2 * j + 1 < p.O2;
        CT acc = CT(0) + y[0][2 * j];
        if (two) acc = acc + y[0][2 * j + 1];
        if (n1 == 2) {
            acc = acc + y[1][2 * j];
            if (two) acc = acc + y[1][2 * j + 1];
        }
        outc.e[j] = narrow<T>(div_count<CT>(acc, n1 * (two ? 2 : 1)));
    }
    S *dst = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + static_cast<int64_t>(pr) * P2 + jo / 2;
    if (jo + E <= p.O2 + 1) {   // all E / 2 pooled elements exist: one store at the element's alignment (pooled rows of any length)
        typedef typename vec_of<8>::type v8 __attribute__((aligned(4)));
        typename vec_of<8>::type bits;
        __builtin_memcpy(&bits, outc.e, 8);
        *reinterpret_cast<v8 *>(dst) = bits;
    } else {
#pragma unroll
        for (int j = 0; j < E / 2; ++j)
            if (jo + 2 * j < p.O2) dst[j] = outc.e[j];
    }
}
template <int ESIZE, int PAD>
__global__ __launch_bounds__(kThreads) void step_gather_forward_small(const GatherParams p) {
    using R_t = typename raw_t<ESIZE>::type;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr int E = 16 / ESIZE;
    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs1, cs2;
    channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2, p.pad);
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int r = step * p.R + tr;
    if (tr >= p.R || r >= p.O1) return;
    const int rb = row_map_t<PAD>(r + p.L1, cs1, p.S1, p.pad);
    const int dcol = p.L2 - cs2;               // source column of output column 0 under the plain shift (uniform)
    const int ph = (dcol * ESIZE) & 15;        // byte phase of every chunk's source window
    const int q = tc + ((dcol * ESIZE) >> 4);  // first aligned source piece of this chunk (may lie outside the row)
    uint32_t fill32 = static_cast<uint32_t>(p.fill) & (ESIZE == 1 ? 0xffu : 0xffffu);
    fill32 = ESIZE == 1 ? fill32 * 0x01010101u : fill32 * 0x00010001u;
    const u4 fillv = {fill32, fill32, fill32, fill32};
    const char *xrow = static_cast<const char *>(p.x) + (static_cast<int64_t>(plane) * p.x_plane + static_cast<int64_t>(rb < 0 ? 0 : rb) * p.S2) * ESIZE;
    R_t *dst = static_cast<R_t *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + r * p.O2 + tc * E;
    const int s0 = tc * E + dcol;
    const bool plain = PAD == 0 || (s0 >= 0 && s0 + E <= p.S2);  // every source column inside the row maps to itself
    u4 o = fillv;
    if (plain) {
        const bool va = rb >= 0 && q >= 0 && q < p.xppr, vb = rb >= 0 && q + 1 >= 0 && q + 1 < p.xppr;
        const int qa = q < 0 ? 0 : (q >= p.xppr ? p.xppr - 1 : q), qb = q + 1 < 0 ? 0 : (q + 1 >= p.xppr ? p.xppr - 1 : q + 1);
        u4 A = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(xrow) + qa);
        u4 B = fillv;
        if (ph != 0) B = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(xrow) + qb);  // uniform
        A = va ? A : fillv;
        B = vb ? B : fillv;
        const uint32_t sh = static_cast<uint32_t>(ph & 3) * 8u;
        switch (ph >> 2) {  // uniform
        case 0: o = u4{__builtin_amdgcn_alignbit(A.y, A.x, sh), __builtin_amdgcn_alignbit(A.z, A.y, sh), __builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh)}; break;
        case 1: o = u4{__builtin_amdgcn_alignbit(A.z, A.y, sh), __builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh)}; break;
        case 2: o = u4{__builtin_amdgcn_alignbit(A.w, A.z, sh), __builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh), __builtin_amdgcn_alignbit(B.z, B.y, sh)}; break;
        default: o = u4{__builtin_amdgcn_alignbit(B.x, A.w, sh), __builtin_amdgcn_alignbit(B.y, B.x, sh), __builtin_amdgcn_alignbit(B.z, B.y, sh), __builtin_amdgcn_alignbit(B.w, B.z, sh)}; break;
        }
    } else if (rb >= 0) {
        // a chunk at a row end under a wrapping / clamping padding: element by element.  (Costs its wave a second memory
        // round trip; issuing these loads unconditionally for every lane -- raw-buffer offsets out of range where not
        // needed -- was measured slower still: C5 reflect 1.52 vs 1.39 ms, against 1.25 ms of plane_gather_forward_lds,
        // which is why the automatic choice takes this kernel for zeros padding only.)
        const R_t *row = reinterpret_cast<const R_t *>(xrow);
        Chunk<R_t, E> v;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int m = row_map_t<PAD>(tc * E + p.L2 + e, cs2, p.S2, p.pad);
            v.e[e] = m >= 0 ? row[m] : static_cast<R_t>(p.fill);
        }
        __builtin_memcpy(&o, v.e, 16);
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(dst));
}

// ---------------------------------------------------------------------------------------------------------------------
// step_forward_lds: forwards that read their source rows through LDS, in the same one-step shape -- the interpolating
// forward of every float dtype (R + 1 corner rows per step) and the sparse-shift forward of 2-byte elements (16-byte
// global loads at 2-byte alignment are slow; aligned LDS-DMA + a funnel shift is not).  No table, no workspace: the
// column state of the thread's chunk is folded arithmetically (one map, the forward has VALU time to spare), the
// channel's two weights come through the scalar cache.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int ND, bool ACTIVE, int PAD, int U>
__global__ __launch_bounds__(kThreads) void step_forward_lds(const FwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int KP = 4;  // staged pieces per thread and plane of the generic (cropped) staging loop
    constexpr int NPL = (ND == 3 && ACTIVE) ? 2 : 1;  // source planes of a step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;  // 64-byte pads in front and behind: see lds_read_row

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spv);
    const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spv);
    const int a = ND == 3 ? static_cast<int>(fdiv(vstep, p.d_spp)) : 0;
    const int step = static_cast<int>(vstep) - a * p.spp;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wv[3];
    load_weights_nd<CT>(p.w, p.wkind, c, p.nd, wv);
    // weights_init_forward (shifts_cuda.cu:168-183): sparse shift rounds (half to even, as the CPU path), active floors
    CT rr[3], dn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rr[d] = ACTIVE ? c_floor<CT>(wv[d]) : c_rint<CT>(wv[d]);
        dn[d] = ACTIVE ? wv[d] - rr[d] : CT(0);
    }
    const int cs0 = ND == 3 ? __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], p.S0, p.d_per0, p.pad)) : 0;
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], p.S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], p.S2, p.d_per2, p.pad));
    // fractions in real-dim order (interp_t's d[]): 2-D (row, inner), 3-D (plane, row, inner)
    const CT dw[3] = {ND == 3 ? dn[0] : dn[1], ND == 3 ? dn[1] : dn[2], ND == 3 ? dn[2] : CT(0)};

    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr, xppr = p.xppr;
    const int RT = U * R;
    const int PR = RT + (ACTIVE ? 1 : 0);   // staged rows per plane
    const int b0 = step * RT;
    const int Rn = min(RT, p.O1 - b0);
    const int last = Rn - (ACTIVE ? 0 : 1);  // last staged row of the step (the interpolating shift: + 1 corner row)
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + static_cast<int64_t>(a) * p.O1 * p.O2;
    int pa[NPL];
#pragma unroll
    for (int h = 0; h < NPL; ++h) pa[h] = ND == 3 ? row_map_t<PAD>(a + p.L0 + h, cs0, S0, p.pad) : 0;

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    auto dma = [&](int src_row, int col_piece, int lds_piece0) {
        const uint32_t off = static_cast<uint32_t>(src_row * S2 + col_piece * E) * static_cast<uint32_t>(sizeof(S));
        char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;  // wave-uniform; hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 2 /* nt */);
    };
    if (xppr == cpr) {  // no crop along the rows: thread (tr, tc) stages piece tc of its own rows (no index arithmetic)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vtr = tr + u * R;
            if (tr < R && vtr <= last) {
                const int src = row_map_t<PAD>(b0 + p.L1 + vtr, cs1, S1, p.pad);
#pragma unroll
                for (int h = 0; h < NPL; ++h)
                    if (src >= 0 && pa[h] >= 0) dma(pa[h] * S1 + src, tc, h * PR * cpr + u * R * cpr);
            }
        }
        if (ACTIVE && Rn == RT && tid < cpr) {
            const int src = row_map_t<PAD>(b0 + p.L1 + RT, cs1, S1, p.pad);
#pragma unroll
            for (int h = 0; h < NPL; ++h)
                if (src >= 0 && pa[h] >= 0) dma(pa[h] * S1 + src, tid, (h * PR + RT) * cpr);
        }
    } else {
        const int npieces = (last + 1) * xppr;
#pragma unroll
        for (int h = 0; h < NPL; ++h) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (k * kThreads < npieces) {  // uniform
                    const int q = k * kThreads + tid;
                    const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_xppr));
                    const int j = q - slot * xppr;
                    int src = row_map_t<PAD>(b0 + p.L1 + slot, cs1, S1, p.pad);
                    if (q >= npieces || pa[h] < 0) src = -1;
                    if (src >= 0) dma(pa[h] * S1 + src, j, h * PR * xppr + k * kThreads);
                }
            }
        }
    }
    const int jo = tc * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {  // zeros: column j reads column j - shift when that is a column (affine everywhere)
        const int base = jo + p.L2 - cs2;
        xm.base = (base + E < 0 || base >= S2) ? 0 : base;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = (base + e >= 0 && base + e < S2) ? base + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(jo + p.L2, cs2, S2, p.pad);
    }
    // window reads: two aligned 16-byte spans and the workgroup's phase (lds_window6: no bank conflicts); chunks that are not
    // affine at that phase (and cropped problems, whose staged rows start at another column) read element by element
    const int phw = ((p.L2 - cs2) * static_cast<int>(sizeof(S))) & 15;
    const bool fastw = xm.affine && ((xm.base * static_cast<int>(sizeof(S))) & 15) == phw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto row_valid = [&](int pr) { return PAD != 0 || row_map_t<PAD>(pr, cs1, S1, p.pad) >= 0; };
    const int RBL = xppr * 16;  // bytes per staged row
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int vtr = tr + u * R;
        if (tr >= R || vtr >= Rn) continue;
        const int b = b0 + vtr;
        Chunk<S, E> res;
        if constexpr (ACTIVE && ND == 3) {
            // the reference nests the blends plane, row, inner (interpolation.h:34-40): blended over the two planes first,
            // a row of E + 1 columns serves the E elements of the chunk -- 3 E + 3 blends per chunk instead of 7 E, same bits
            CT rowb[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                S r0[E + 1], r1[E + 1];
                const bool rv = row_valid(b + p.L1 + hb);
                lds_read_row_span<S, E>(tile + (vtr + hb) * RBL, rv && pa[0] >= 0, xm, fastw, phw, r0);
                lds_read_row_span<S, E>(tile + (PR + vtr + hb) * RBL, rv && pa[1] >= 0, xm, fastw, phw, r1);
#pragma unroll
                for (int e = 0; e <= E; ++e) {
                    const CT two[2] = {widen<T>(r0[e]), widen<T>(r1[e])};
                    rowb[hb][e] = interp_t<T, 1>(two, &dw[0]);
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
        } else if constexpr (ACTIVE) {
            CT xv[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                S raw[E + 1];
                lds_read_row_span<S, E>(tile + (vtr + hb) * RBL, row_valid(b + p.L1 + hb), xm, fastw, phw, raw);
#pragma unroll
                for (int e = 0; e <= E; ++e) xv[hb][e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
                res.e[e] = narrow<T>(interp_t<T, 2>(v, dw));
            }
        } else {
            S raw[E + 1], fill;
            const typename raw_t<sizeof(S)>::type fill_bits = static_cast<typename raw_t<sizeof(S)>::type>(p.fill);
            __builtin_memcpy(&fill, &fill_bits, sizeof(S));
            const bool valid = pa[0] >= 0 && row_valid(b + p.L1);
            lds_read_row_span<S, E>(tile + vtr * RBL, valid, xm, fastw, phw, raw);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = (valid && xm.cm[e] >= 0) ? raw[e] : fill;
        }
        store_chunk<S, E>(op + static_cast<int64_t>(b) * p.O2 + jo, res);
    }
}

}  // namespace

// sparse-shift / quantized forward of 4- and 8-byte elements: dense tensors, output rows of whole 16-byte chunks and at
// most one workgroup pass wide (crops are fine: a gather)
bool step_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[1] == 1) return false;
    const int es = dtype_size(dtype);
    if (es != 1 && es != 2 && es != 4 && es != 8) return false;
    // (the interpolating shift: step_forward_lds / span_forward.  Its direct-load form -- every corner row loaded by two workgroups at
    //  element alignment -- measured 1.51 vs 1.13 ms on the C2 tensor in round 3 and was removed in round 4)
    if (g.active && dtype <= SHIFTND_BF16) return false;
    // 2-D; 3-D for the sparse shift of 4- / 8-byte elements (the caller checks that its weights are floats)
    if (g.nd == 3 ? es < 4 : (g.nd != 2 || g.S[0] != 1 || g.O[0] != 1)) return false;
    // 1- / 2-byte elements: aligned pieces of the source rows
    if (es < 4 && ((g.S[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0)) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe < 1 || oe < 1 || xe >= (1LL << 30) || oe >= (1LL << 30)) return false;
    if ((g.O[2] * es) % 16 != 0 || g.O[2] * es / 16 > kThreads || reinterpret_cast<uintptr_t>(out) % 16 != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    const int cpr = static_cast<int>(g.O[2] * es / 16);
    const int64_t R = kThreads / cpr;
    const int64_t spp = (g.O[1] + R - 1) / R;
    if (g.N * g.C * g.O[0] * spp + 8 >= (1LL << 31)) return false;
    if (g_step_tune[1] == 2) return true;
    // 1- / 2-byte elements: zeros padding only (row-end chunks of the other paddings go element by element), planes of at
    // least 16 KiB (2-byte) / 32 KiB (1-byte): below that the per-channel kernels that walk many planes win
    const int64_t pe = g.O[1] * g.O[2];   // (a plane's elements: a 3-D volume of small planes is no better off)
    if (es < 4) return g.pad == 0 && pe * es >= (es == 2 ? 16 : 32) * 1024;
    return pe * es >= 32 * 1024;  // as the sweep kernels: small planes go to the per-channel walk
}

int step_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                 void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    GatherParams p{};
    p.pad = g.pad;
    p.x = x;
    p.out = out;
    p.w = w;
    p.wzp = wzp;
    p.fill = fill_bits;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.x_plane = g.S[1] * g.S[2];
    p.o_plane = g.O[1] * g.O[2];
    p.cpr = static_cast<int>(g.O[2] * es / 16);
    p.xppr = static_cast<int>(g.S[2] * es / 16);
    p.R = kThreads / p.cpr;
    if (p.R > p.O1) p.R = p.O1;
    p.spp = (p.O1 + p.R - 1) / p.R;
    p.S0 = static_cast<int>(g.S[0]);
    p.O0 = static_cast<int>(g.O[0]);
    p.L0 = static_cast<int>(g.L[0]);
    p.spv = p.O0 * p.spp;
    if (g.nd == 3) {
        p.x_plane = g.S[0] * g.S[1] * g.S[2];
        p.o_plane = g.O[0] * g.O[1] * g.O[2];
    }
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spv;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_spv = make_fastdiv(static_cast<uint32_t>(p.spv));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(p.cpr));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    if (g.nd == 3) {   // sparse shift of 4- / 8-byte elements, float weights (the eligibility check and the caller see to that)
        note_kernel("step_gather_forward");
#define SHIFTND_STEP_FWD3(ES) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((step_gather_forward<ES, 0, 3>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((step_gather_forward<ES, 1, 3>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((step_gather_forward<ES, 2, 3>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((step_gather_forward<ES, kPadMirror, 3>), grid, block, 0, st, p); break; \
    }
        if (es == 4) { SHIFTND_STEP_FWD3(4) } else { SHIFTND_STEP_FWD3(8) }
#undef SHIFTND_STEP_FWD3
        return SHIFTND_OK;
    }
    note_kernel(es < 4 ? "step_gather_forward_small" : "step_gather_forward");
#define SHIFTND_STEP_FWD(KERNEL, ES) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((KERNEL<ES, 0>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((KERNEL<ES, 1>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((KERNEL<ES, 2>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((KERNEL<ES, kPadMirror>), grid, block, 0, st, p); break;   /* reflect and symmetric */ \
    }
    if (es == 1) { SHIFTND_STEP_FWD(step_gather_forward_small, 1) }
    else if (es == 2) { SHIFTND_STEP_FWD(step_gather_forward_small, 2) }
    else if (es == 4) { SHIFTND_STEP_FWD(step_gather_forward, 4) }
    else { SHIFTND_STEP_FWD(step_gather_forward, 8) }
#undef SHIFTND_STEP_FWD
    return SHIFTND_OK;
}


// interpolating forward of every float dtype, sparse-shift forward of 2-byte elements: dense 2-D tensors, source rows and
// output rows of whole 16-byte pieces, at most one workgroup pass wide
bool step_forward_lds_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[2] == 1) return false;
    if (dtype > SHIFTND_BF16) return false;
    const int es = dtype_size(dtype);
    const bool interpolating = g.active != 0;
    if (!interpolating && es != 2) return false;
    if ((g.nd != 2 && g.nd != 3) || (g.nd == 2 && (g.S[0] != 1 || g.O[0] != 1))) return false;
    // 3-D: 4- / 8-byte interpolation (N8 C128 16x112x112 fp32 0.38 -> 0.30 ms) and the 2-byte sparse shift (0.155 -> 0.136 ms);
    // 16-bit interpolation is walk_forward16's (this kernel measured 0.215 ms against the sliding window's 0.18: four corner rows
    // to unpack per output row against two -- the form is no longer built)
    if (g.nd == 3 && interpolating && es == 2) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe < 1 || oe < 1 || xe >= (1LL << 30) || oe >= (1LL << 30) || g.S[2] > 32000) return false;
    if ((g.S[2] * es) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return false;
    if ((g.O[2] * es) % 16 != 0 || g.O[2] * es / 16 > kThreads || reinterpret_cast<uintptr_t>(out) % 16 != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    const int cpr = static_cast<int>(g.O[2] * es / 16), xppr = static_cast<int>(g.S[2] * es / 16);
    int64_t R = kThreads / cpr;
    if (R > g.O[1]) R = g.O[1];
    const int64_t npl = (g.nd == 3 && interpolating) ? 2 : 1;
    if ((2 * R + 1) * xppr > 4 * kThreads) return false;  // (heavy crops: few output chunks per source row)
    if (64 + npl * (2 * R + 1) * xppr * 16 + 64 > 64 * 1024) return false;
    const int64_t spp = (g.O[1] + R - 1) / R;
    if (g.N * g.C * g.O[0] * spp + 8 >= (1LL << 31)) return false;
    // knob 34: 0 = automatic (two row groups per thread), 1 = never, 2 / 3 = always, with one / two row groups
    if (g_step_tune[2] >= 2) return true;
    // same box, per-channel LDS kernels -> this one: interpolating fp32 N64 C256 224x224 1.09 -> 1.00 ms, N16 C64 448x448
    // 0.39 -> 0.26, N128 C256 56x56 0.158 -> 0.135, bf16 0.079 -> 0.068, N256 C512 8x8 0.103 -> 0.086; sparse fp16 reflect
    // C512 224x224 1.21 -> 1.06, bf16 56x56 0.078 -> 0.064; the one loss: sparse 2-byte planes of 2 KiB (32x32: 0.059 -> 0.064)
    return interpolating || g.O[1] * g.O[2] * es >= 4 * 1024;
}

// (only the forms a call can reach are instantiated: the sparse shift as one raw 2-byte copy, the 3-D interpolation for 4- / 8-byte
// elements -- walk_forward16 serves the 16-bit volumes)
template <typename T, bool ACT>
static void launch_step_forward_lds(const FwdParams &p, int pad, size_t lds, hipStream_t st) {
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    constexpr bool k3d = !ACT || sizeof(typename T::S) >= 4;
#define SHIFTND_STEP_FWD_LDS(PADV) \
    case PADV: \
        if (p.nd == 3) { \
            if constexpr (k3d) hipLaunchKernelGGL((step_forward_lds<T, 3, ACT, PADV, 2>), grid, block, lds, st, p); \
        } else hipLaunchKernelGGL((step_forward_lds<T, 2, ACT, PADV, 2>), grid, block, lds, st, p); \
        break;
    switch (pad) { SHIFTND_STEP_FWD_LDS(0) SHIFTND_STEP_FWD_LDS(1) SHIFTND_STEP_FWD_LDS(2) default: SHIFTND_STEP_FWD_LDS(3) }
#undef SHIFTND_STEP_FWD_LDS
}

int step_forward_lds(const Geometry &g, int dtype, const void *x, const void *w, int wkind, uint64_t fill_bits, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    FwdParams p{};
    p.pad = g.pad;
    p.x = x;
    p.out = out;
    p.w = w;
    p.fill = fill_bits;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.S0 = static_cast<int>(g.S[0]);
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O0 = static_cast<int>(g.O[0]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L0 = static_cast<int>(g.L[0]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.o_plane = g.O[0] * g.O[1] * g.O[2];
    p.cpr = static_cast<int>(g.O[2] * es / 16);
    p.xppr = static_cast<int>(g.S[2] * es / 16);
    p.R = kThreads / p.cpr;
    if (p.R > p.O1) p.R = p.O1;
    // two row groups per thread (one row group: C2-tensor interpolating forward 1.15 ms, two: 1.00 ms); only that form is built -- a
    // plane of fewer rows than one group is a ragged last step like any other (rows beyond the plane stage and store nothing)
    const int U = 2;
    p.spp = (p.O1 + U * p.R - 1) / (U * p.R);
    p.spv = p.O0 * p.spp;
    p.d_spv = make_fastdiv(static_cast<uint32_t>(p.spv));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spv;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(p.cpr));
    p.d_xppr = make_fastdiv(static_cast<uint32_t>(p.xppr));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const bool active = g.active != 0;
    const size_t lds = 64 + static_cast<size_t>((g.nd == 3 && active) ? 2 : 1) * (U * p.R + (active ? 1 : 0)) * p.xppr * 16 + 64;
    note_kernel(active ? "step_active_forward" : "step_gather_forward_lds");
    if (!active) {  // a raw copy of 2-byte elements: one instantiation serves fp16 and bf16
        launch_step_forward_lds<f16_t, false>(p, g.pad, lds, st);
        return SHIFTND_OK;
    }
    switch (dtype) {
    case SHIFTND_F32: launch_step_forward_lds<f32_t, true>(p, g.pad, lds, st); break;
    case SHIFTND_F64: launch_step_forward_lds<f64_t, true>(p, g.pad, lds, st); break;
    case SHIFTND_F16: launch_step_forward_lds<f16_t, true>(p, g.pad, lds, st); break;
    default: launch_step_forward_lds<bf16_t, true>(p, g.pad, lds, st); break;
    }
    return SHIFTND_OK;
}

// the 2-D sparse shift + 2 x 2 average pool of 4- / 8-byte float elements in one sweep of one-step workgroups
bool step_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[1] == 1) return false;
    // (round 6: 16-bit types too -- fp32 sums like ATen's, one rounding; knob 33 = 3 keeps them on the band-walk kernel)
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F64 && !((dtype == SHIFTND_F16 || dtype == SHIFTND_BF16) && g_step_tune[1] != 3)) return false;
    if (g.nd != 2 || g.K[1] != 2 || g.K[2] != 2 || g.S[0] != 1 || g.O[0] != 1) return false;
    // (round 6: the interpolating shift too -- 4- / 8-byte elements: N32 C256 112x112 fp32 0.183 -> 0.148 ms; bf16, whose element-aligned
    //  16-byte loads are slow, 0.131 -> 0.163: those keep the band-walk kernel.  Knob 33 = 4: not)
    if (g.active && (g_step_tune[1] == 4 || g.S[1] < 2 || g.S[2] < 2 || dtype_size(dtype) < 4)) return false;
    const int es = dtype_size(dtype);
    const int64_t xe = g.S[1] * g.S[2], oe = g.O[1] * g.O[2];
    if (xe < 1 || oe < 1 || xe >= (1LL << 30) || oe >= (1LL << 30)) return false;
    // (round 6: windows of any width -- the last chunk of a row may be partial; pooled rows at the element's alignment)
    if ((g.O[2] * es + 15) / 16 > kThreads || reinterpret_cast<uintptr_t>(out) % es != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S)) return false;
    const int64_t cpr = (g.O[2] * es + 15) / 16, R = kThreads / cpr, p1 = (g.O[1] + 1) / 2;
    return g.N * g.C * ((p1 + R - 1) / R) + 8 < (1LL << 31);
}

int step_forward_pooled(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    GatherParams p{};
    p.pad = g.pad;
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.x_plane = g.S[1] * g.S[2];
    p.o_plane = g.P[1] * g.P[2];   // (the pooled plane)
    p.cpr = static_cast<int>((g.O[2] * es + 15) / 16);
    // (16-bit: the aligned-pieces form needs source rows of whole pieces at a 16-byte boundary; 0: element-aligned loads)
    p.xppr = ((g.S[2] * es) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0) ? static_cast<int>(g.S[2] * es / 16) : 0;
    const int P1 = static_cast<int>(g.P[1]);
    p.R = kThreads / p.cpr;
    if (p.R > P1) p.R = P1;
    p.spp = (P1 + p.R - 1) / p.R;
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(p.cpr));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    note_kernel("step_gather_forward_pool");
#define SHIFTND_STEP_FWD_POOL_A(TT, ACT) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((step_gather_forward_pool<TT, 0, ACT>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((step_gather_forward_pool<TT, 1, ACT>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((step_gather_forward_pool<TT, 2, ACT>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((step_gather_forward_pool<TT, kPadMirror, ACT>), grid, block, 0, st, p); break; \
    }
#define SHIFTND_STEP_FWD_POOL(TT) \
    if (g.active) { SHIFTND_STEP_FWD_POOL_A(TT, true) } else { SHIFTND_STEP_FWD_POOL_A(TT, false) }
    if (dtype == SHIFTND_F32) { SHIFTND_STEP_FWD_POOL(f32_t) }
    else if (dtype == SHIFTND_F64) { SHIFTND_STEP_FWD_POOL(f64_t) }
    else if (dtype == SHIFTND_F16) { SHIFTND_STEP_FWD_POOL_A(f16_t, false) }   // (16-bit: the sparse shift only, step_forward_pooled_eligible)
    else { SHIFTND_STEP_FWD_POOL_A(bf16_t, false) }
#undef SHIFTND_STEP_FWD_POOL_A
#undef SHIFTND_STEP_FWD_POOL
    return SHIFTND_OK;
}

}  // namespace shiftnd
