// shiftnd_walk.hip -- the 3-D backward of 16-bit tensors as a walk through the planes, round 4 (gfx950 / MI355X).
// DESIGN section 3.17.  BASELINE config 3 (Shift3d active, N8 C128 16x112x112 bf16) runs here.
//
// The walk itself is round 3's (shiftnd_step.hip, walk_backward): a workgroup owns R consecutive rows of one (n, c) volume
// and steps through its planes; per step ONE plane of the saved input and ONE of the incoming gradient are staged (R + 1
// rows each, global -> registers -> LDS, two planes ahead), the "+0" corner plane of a step is the previous step's "+1"
// plane and stays in registers.  That kernel moved exactly the algorithmic bytes but issued ~200 vector instructions per
// wave and step from 138 (zeros padding) to 192 (the other paddings) VGPRs: 3 and 2 waves per SIMD.  What changed:
//
//   * LDS layout by DWORD PLANES: dword i of piece p lives at plane[i][p].  The 5 dwords of a thread's shifted window are
//     5 ds_read_b32 whose lanes hit consecutive banks (no conflicts), and the window's phase is part of the ADDRESS --
//     no two-span read + uniform switch + 24 register moves per step (lds_window6), and no funnel shift per window.
//   * two zero pieces behind every staged row (zeros padding): a window that leaves its row reads zeros -- no column masks
//     (36 v_and per step).  The other paddings never mask: their maps have a source for every column.
//   * the half-word parity of a window is handled where it is cheapest: the weight-gradient sums pair the x dwords AS THEY
//     LIE with the own gradient chunk either as it lies or shifted by one element (5 shifts per step instead of 16 + 16),
//     the blends unpack the right halves (two copies of the two sections, chosen by a uniform branch).
//   * 16-bit interpolation nests the three blends inner-first (row, column, then plane) instead of the reference's plane,
//     row, column (interpolation.h:34-40): the row / column blend of a plane is computed once and carried to the next step
//     -- 8 fp32 values instead of 18, 50 instead of 70 blend instructions.  fp32 / fp64 tensors keep the reference's nesting
//     bit for bit (shiftnd_step.hip); 16-bit tensors have no executable reference arithmetic (fp32 math on the
//     widened inputs, one rounding, within 1 ulp of the 16-bit type) and the tests hold this kernel to exactly that.
//   * the running sums stay fp32 in registers across at most 16 planes and are flushed per WAVE (DPP tree) into fp64
//     slots: no per-thread fp64 slots in LDS (16 KB), 16.6 KB of LDS per workgroup in all.
//   * chunks whose column map is not a plain shift (the row ends of border / periodic / reflect / symmetric) gather their
//     9 elements through 9 precomputed LDS addresses; the padding mode is a run-time value there (two instantiations per
//     dtype and shift kind instead of five).
//
// Reference behaviour restated: kernels/shifts_kernels.h:222-327 (backward), :132-154 (weight gradients),
// kernels/interpolation.h:34-61; cpu/shifts_cpu.cpp:242-244 (weight preparation: step_prep).
#include "shiftnd_step.hpp"

namespace shiftnd {
namespace {

constexpr int kWalkSlots = 512;                        // piece slots per dword plane: 448 of the tile, 64 dump slots
constexpr int kWalkPlaneBytes = kWalkSlots * 4;
constexpr int kWalkTileBytes = 4 * kWalkPlaneBytes;    // one tensor's tile
constexpr int kWalkMargin = 2;                         // zero pieces in front of row 0
constexpr int kWalkGuard = 2;                          // zero pieces behind every row
constexpr int kWalkDump0 = 448;
constexpr int kWalkSmall = 6;                          // |column shift| up to which the guards hold the folded row ends (paddings 1 .. 4)
constexpr int kWalkFlush = 16;                         // planes between two flushes of the fp32 sums

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

// The workgroup barrier of the walk kernels.  (Round 4 wrapped it in an explicit lgkmcnt wait and compiler barriers against what
// looked like an LDS race of walk_backward16<..., ZEROS = false>; round 5 found the cause elsewhere -- a store-data hazard behind
// buffer_store_dwordx4 with an SGPR soffset, shiftnd_common.hpp: buffer_store_b128_soffset -- and the barrier is plain again: every
// s_barrier of these kernels is preceded by s_waitcnt lgkmcnt(0) with no LDS instruction in between, tools/isa_barriers.py.)
__device__ __forceinline__ void walk_barrier() { __syncthreads(); }

// cache-policy bits: the staged loads and the own chunk are plain (every gradient plane is read twice by its workgroup, the
// "+1" corner row by two workgroups: nontemporal loads cost 20 %), grad_x is written once and not read again: nontemporal
// (C3 0.243 -> 0.236 ms).  Measured and dropped: walks of 8 / 4 / 2 / 1 planes per workgroup, an occupancy
// limit, plain dispatch order instead of XCD-contiguous ids, no "+1" corner row -- the memory skeleton of the walk (loads and
// stores only) stays at 0.22 - 0.24 ms in every form; the kernel runs within 3 % of that skeleton.
constexpr int kWalkStoreAux = 2;

// byte offset (within a tile) of dword D of the piece-linear dword stream
__device__ __forceinline__ uint32_t walk_dword_at(int D) { return static_cast<uint32_t>((D & 3) * kWalkPlaneBytes + (D >> 2) * 4); }
// ... of 16-bit element m of the row whose first piece sits in slot `slot0`
__device__ __forceinline__ uint32_t walk_elem_at(int slot0, int m) {
    return static_cast<uint32_t>(((m >> 1) & 3) * kWalkPlaneBytes + (slot0 + (m >> 3)) * 4 + (m & 1) * 2);
}

template <typename T> __device__ __forceinline__ float half_value(uint32_t dword, int hi) {
    const uint16_t h = static_cast<uint16_t>(hi ? dword >> 16 : dword);
    return widen<T>(__builtin_bit_cast(typename T::S, h));
}

// Column state of a thread's window through one map.  NA addresses: zeros padding 5 (the window's dwords); otherwise 9 --
// a plain-shift chunk uses the first 5 as dword addresses, any other chunk all 9 as element addresses.
//
// Paddings 1 .. 4 with |shift| <= 6 (launch-uniform `small`: weights start in (-1, 1) and stay small): the window of a row-end
// chunk reaches at most seven elements in front of its row and eight behind it -- the row's two guard pieces.  Every element there
// is a fold of the row's FIRST or LAST piece (border: one element repeated; periodic: the other end's piece as it is; symmetric:
// the piece reversed; reflect: reversed and moved by one element), and the threads that stage those two pieces hold them in
// registers: they park a second, permuted copy into the guards (walk_park_guards), so EVERY chunk reads its window as five plain
// dwords, exactly like the zeros-padding kernel.  (Round 4 read the window between ZERO guards and or-ed up to two folded elements
// in per read for |shift| <= 1 -- 16 VGPRs of masks and addresses -- and every other shift made each wave, all of which hold some
// row-end chunks, run the 9-address gather beside the plain path: 3 x the LDS instructions of the zeros-padding kernel.  Measured
// and dropped in round 4: launch-uniform per-read fix-ups for |shift| <= 7 -- 127 -> 187 VGPRs.)
template <int NA> struct WalkWindow {
    uint32_t at[NA];   // byte offsets within the tile, row 0 of the thread (row 1: + row pitch)
    bool plain;        // the window is 5 consecutive dwords
};

// The guard pieces of the row whose first piece sits in slot `slot0`: slot0 - 1 (elements -8 .. -1) and slot0 + cpr (elements
// S2 .. S2 + 7).  A thread that holds the row's first / last piece `v` (v.x = elements 0, 1 ... v.w = elements 6, 7 of the piece)
// parks the fold of it:
//   left guard  (from the FIRST piece f; periodic: the LAST piece as it is)
//       border     f0 f0 f0 f0 f0 f0 f0 f0          symmetric  f7 f6 f5 f4 f3 f2 f1 f0        reflect  -- f7 f6 f5 f4 f3 f2 f1
//   right guard (from the LAST piece l; periodic: the FIRST piece as it is)
//       border     l7 l7 l7 l7 l7 l7 l7 l7          symmetric  l7 l6 l5 l4 l3 l2 l1 l0        reflect  l6 l5 l4 l3 l2 l1 l0 --
// (--: element -8 / S2 + 7, which no window of |shift| <= 6 reads.)
struct WalkGuards { uint32_t at; bool left, on; };
__device__ __forceinline__ WalkGuards walk_guards(bool fill, bool own, bool first, bool last, int pad, int slot0, int cpr, int lane) {
    const bool wrap = pad == 2;   // periodic: the guards come from the other end of the row
    WalkGuards g;
    g.left = wrap ? last : first;
    g.on = fill && own && (first || last);
    g.at = static_cast<uint32_t>(g.on ? (g.left ? slot0 - 1 : slot0 + cpr) : kWalkDump0 + lane) * 4u;
    return g;
}
__device__ __forceinline__ void walk_park_guards(char *tile, const WalkGuards &g, const u4_t &v, int pad) {
    if (!g.on) return;   // (two lanes in cpr hold a row end; masked: same-box A / B against unconditional stores to dump slots, C3 backward 0.250 -> 0.246 ms)
    auto rot = [](uint32_t a) { return __builtin_amdgcn_alignbit(a, a, 16); };   // swap the halves
    const u4_t rev = u4_t{rot(v.w), rot(v.z), rot(v.y), rot(v.x)};                // e7 e6 | e5 e4 | e3 e2 | e1 e0
    u4_t o;
    if (pad == 2) {          // (uniform)
        o = v;
    } else if (pad == 4) {
        o = rev;
    } else if (pad == 3) {   // the reversed piece moved by one element: towards the row on either side
        const u4_t l = u4_t{rev.x << 16, __builtin_amdgcn_alignbit(rev.y, rev.x, 16), __builtin_amdgcn_alignbit(rev.z, rev.y, 16),
                            __builtin_amdgcn_alignbit(rev.w, rev.z, 16)};                      // -- e7 | e6 e5 | e4 e3 | e2 e1
        const u4_t r = u4_t{__builtin_amdgcn_alignbit(rev.y, rev.x, 16), __builtin_amdgcn_alignbit(rev.z, rev.y, 16),
                            __builtin_amdgcn_alignbit(rev.w, rev.z, 16), rev.w >> 16};         // e6 e5 | e4 e3 | e2 e1 | e0 --
        o = g.left ? l : r;
    } else {                 // border: the row's first / last element
        const uint32_t e = g.left ? (v.x & 0xffffu) : (v.w >> 16);
        const uint32_t ee = e | (e << 16);
        o = u4_t{ee, ee, ee, ee};
    }
    uint32_t *q = reinterpret_cast<uint32_t *>(tile + g.at);
    q[0] = o.x;
    q[kWalkSlots] = o.y;
    q[2 * kWalkSlots] = o.z;
    q[3 * kWalkSlots] = o.w;
}

template <bool ZEROS, int NA, bool LEAN = false>
__device__ __forceinline__ WalkWindow<NA> walk_window(int ji, int cs, int S2, int pad, int slot0, bool live, bool small) {
    WalkWindow<NA> w;
    w.plain = true;
    int D = 0;   // the margin: zeros
    const int first = ji - cs;   // column of window element 0
    if constexpr (ZEROS) {
        if (live && first + 8 >= 0 && first < S2) D = slot0 * 4 + ((first * 2) >> 2);   // (floor: first >= -8)
#pragma unroll
        for (int i = 0; i < NA; ++i) w.at[i] = walk_dword_at(D + i);
    } else if (LEAN || small) {   // (uniform; cs is the SIGNED shift here, |cs| <= 6, and the row has at least two chunks: the guards hold
                          //  the folded elements -- walk_park_guards -- and the window is five plain dwords)
        if (live) D = slot0 * 4 + ((first * 2) >> 2);
#pragma unroll
        for (int i = 0; i < NA; ++i) w.at[i] = walk_dword_at(D + (i < 5 ? i : 4));
    } else {
        int cm[9];
        bool run = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            cm[k] = fold_index(ji + k - cs, S2, pad);   // paddings 1 .. 4: always a column
            run = run && cm[k] == cm[0] + k;
        }
        w.plain = run || !live;
        if (live && run) D = slot0 * 4 + (cm[0] >> 1);
        if (w.plain) {
#pragma unroll
            for (int i = 0; i < NA; ++i) w.at[i] = walk_dword_at(D + (i < 5 ? i : 4));
        } else {
#pragma unroll
            for (int k = 0; k < NA; ++k) w.at[k] = walk_elem_at(slot0, cm[k < 9 ? k : 8]);
        }
    }
    return w;
}

// the 5 dwords of a window: half (k + PAR) of the result is window element k
template <bool ZEROS, int NA, int PAR>
__device__ __forceinline__ void walk_read(const char *tile, const WalkWindow<NA> &w, uint32_t row_off, uint32_t (&o)[5]) {
    if (ZEROS || w.plain) {
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = *reinterpret_cast<const uint32_t *>(tile + w.at[i] + row_off);
    } else {
        uint32_t h[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = *reinterpret_cast<const uint16_t *>(tile + w.at[k < NA ? k : 0] + row_off);
        if constexpr (PAR == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = h[2 * i] | (h[2 * i + 1] << 16);
            o[4] = h[8];
        } else {
            o[0] = h[0] << 16;
#pragma unroll
            for (int i = 1; i < 5; ++i) o[i] = h[2 * i - 1] | (h[2 * i] << 16);
        }
    }
}

// LEAN (paddings 1 .. 4): the workgroup's channel has small column shifts in both maps -- every window is five plain dwords between
// filled guards, no gather path, five window addresses instead of nine.  The kernel picks the body per workgroup (its channel's shifts).
//
// CROP (round 6, zeros padding only): the incoming gradient is the gradient of a WINDOW [O0, O1, O2] at (L0, L1, L2) of the volume
// (ops/shifts.cpp:93-135; Shift3d behind emulate_dw).  The gradient tile holds window rows as they lie from window column 0 -- piece tc
// = window columns 8 tc .. 8 tc + 7, masked beyond O2 (the load runs on into the next row) -- so the crop along the row is one more
// column shift of the gradient window (cg2 + L2), along rows and planes an offset of the staged row / plane index; the own chunk
// (window columns ji - L2 ..) comes from the two aligned pieces around it through a funnel of L2 <= 2 elements, and grad_x is masked to
// the window (the reference leaves it zero outside: shifts_kernels.h:291-313 with the window's sizes).  Window rows of an even number
// of elements (every row starts on a dword), host: walk16_crop_geometry_ok.
template <typename T, bool ACTIVE, bool ZEROS, bool LEAN, bool CROP = false>
__device__ __forceinline__ void walk_backward16_body(const StepParams &p) {
    using S = typename T::S;
    static_assert(sizeof(S) == 2, "16-bit element types");
    static_assert(!CROP || (ZEROS && !LEAN), "the cropped walk: zeros padding");
    constexpr int E = 8;
    constexpr bool PLAIN = ZEROS || LEAN;   // (what walk_read needs to know)
    constexpr int NA = PLAIN ? 5 : 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const tx = smem;
    char *const tg = smem + kWalkTileBytes;
    double *const wsum = reinterpret_cast<double *>(smem + 2 * kWalkTileBytes);   // [waves][8]

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);   // XCD-contiguous ids
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c)
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    constexpr int a0 = 0;   // the planes this workgroup walks through: all of them
    const int a1 = p.S0;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int pad = ZEROS ? 0 : p.pad;
    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr;
    const int b0 = step * R;
    const int Rn = min(R, S1 - b0);
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    const int ji = tc * E;
    const int RP = cpr + kWalkGuard;   // row pitch of the tile in pieces

    // ---- zero the tiles (margin, guards) and the wave sums ------------------------------------------------------------------
    for (int o = tid * 16; o < 2 * kWalkTileBytes + (kThreads / 64) * 8 * static_cast<int>(sizeof(double)); o += kThreads * 16)
        *reinterpret_cast<u4_t *>(__builtin_assume_aligned(smem + o, 16)) = u4_t{0u, 0u, 0u, 0u};

    // ---- the pieces this thread stages, the same for every plane: piece tc of row tr, tr <= min(R, Rn) ----------------------
    // the gradient's geometry: the volume's own, or (CROP) the window's (StepParams wO*, wL*)
    const int O0 = CROP ? p.wO0 : S0, O1 = CROP ? p.wO1 : S1, O2 = CROP ? p.wO2 : S2;
    const int L0 = CROP ? p.wL0 : 0, L1 = CROP ? p.wL1 : 0, L2 = CROP ? p.wL2 : 0;
    const bool own = tr <= R && tr <= Rn;
    const int sx_own = own ? row_map(b0 + tr, d.cx1, S1, pad) : -1;
    const int sg_own = (own && (ACTIVE || tr < R)) ? row_map(b0 + tr - L1, d.cg1, O1, pad) : -1;
    constexpr uint32_t kOOR = 0x80000000u;
    constexpr int kRsrcFlags = 0x00020000;
    const uint32_t plane_bytes = static_cast<uint32_t>(S1) * static_cast<uint32_t>(S2) * 2u;
    const uint32_t vol_bytes = static_cast<uint32_t>(S0) * plane_bytes;   // < 2^31 (host)
    const uint32_t gplane_bytes = CROP ? static_cast<uint32_t>(O1) * static_cast<uint32_t>(O2) * 2u : plane_bytes;
    const uint32_t gvol_bytes = CROP ? static_cast<uint32_t>(O0) * gplane_bytes : vol_bytes;
    const char *xp = reinterpret_cast<const char *>(static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane);
    const char *gp = reinterpret_cast<const char *>(static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * (CROP ? p.g_plane : p.x_plane));
    char *gxp = reinterpret_cast<char *>(static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, vol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gp), 0, gvol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(gxp, 0, vol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, 0, kRsrcFlags);
    const uint32_t vx_own = sx_own >= 0 ? static_cast<uint32_t>(sx_own * S2 + ji) * 2u : kOOR;
    const uint32_t vg_own = sg_own >= 0 ? static_cast<uint32_t>(sg_own * O2 + ji) * 2u : kOOR;   // (CROP: window columns ji .., as they lie)
    // CROP: dword masks of a piece -- the staged gradient piece keeps window columns ji + k < O2; the own chunk and grad_x keep the
    // columns ji + k of the volume that lie in the window's row (and nothing in a row outside the window)
    auto dword_masks = [&](int first, int len, bool row_in, uint32_t (&m)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool lo = row_in && static_cast<unsigned>(first + 2 * i) < static_cast<unsigned>(len);
            const bool hi = row_in && static_cast<unsigned>(first + 2 * i + 1) < static_cast<unsigned>(len);
            m[i] = (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
        }
    };
    uint32_t mstage[4] = {~0u, ~0u, ~0u, ~0u}, mown[4] = {~0u, ~0u, ~0u, ~0u};
    if constexpr (CROP) {
        dword_masks(ji, O2, true, mstage);
        dword_masks(ji - L2, O2, static_cast<unsigned>(b0 + tr - L1) < static_cast<unsigned>(O1), mown);
    }
    // (a thread without a piece loads zeros -- out-of-range offset -- and parks them in a dump slot: every memory
    //  instruction of the loop is unconditional, the compiler's wait counts are exact)
    const uint32_t park_at = static_cast<uint32_t>(own ? kWalkMargin + tr * RP + tc : kWalkDump0 + (tid & 63)) * 4u;
    struct Staged { u4_t xo, go; };
    auto load_planes = [&](int pax, int pag, Staged &v) {   // source planes (uniform; -1: fill)
        v.xo = __builtin_amdgcn_raw_buffer_load_b128(pax >= 0 ? xres : none, vx_own, pax >= 0 ? static_cast<uint32_t>(pax) * plane_bytes : 0u, 0);
        v.go = __builtin_amdgcn_raw_buffer_load_b128(pag >= 0 ? gres : none, vg_own, pag >= 0 ? static_cast<uint32_t>(pag) * gplane_bytes : 0u, 0);
    };
    // the plane maps: x over the volume; the gradient over the volume or (CROP) over the window, whose plane 0 is the volume's L0
    auto xmap0 = [&](int a) { return row_map(a, d.cx0, S0, pad); };
    auto gmap0 = [&](int a) { return row_map(a - L0, d.cg0, O0, pad); };
    auto park_piece = [&](char *tile, const u4_t &v) {
        uint32_t *q = reinterpret_cast<uint32_t *>(tile + park_at);
        q[0] = v.x;
        q[kWalkSlots] = v.y;
        q[2 * kWalkSlots] = v.z;
        q[3 * kWalkSlots] = v.w;
    };
    const int slot0 = kWalkMargin + tr * RP;
    // (the signed column shifts; |shift| <= 6 and at least two chunks per row: the folded row ends live in the guards, see WalkWindow)
    const int per2 = map_period(S2, pad);
    const int sx2 = (!ZEROS && per2 && 2 * d.cx2 > per2) ? d.cx2 - per2 : d.cx2, sg2 = (!ZEROS && per2 && 2 * d.cg2 > per2) ? d.cg2 - per2 : d.cg2;
    const bool small_x = LEAN || (!ZEROS && cpr >= 2 && sx2 >= -kWalkSmall && sx2 <= kWalkSmall), small_g = LEAN || (!ZEROS && cpr >= 2 && sg2 >= -kWalkSmall && sg2 <= kWalkSmall);
    const WalkGuards gdx = walk_guards(small_x, own, tc == 0, tc == cpr - 1, pad, slot0, cpr, tid & 63);
    const WalkGuards gdg = walk_guards(small_g, own && (ACTIVE || tr < R), tc == 0, tc == cpr - 1, pad, slot0, cpr, tid & 63);
    auto park = [&](const Staged &v) {
        park_piece(tx, v.xo);
        if constexpr (CROP) park_piece(tg, u4_t{v.go.x & mstage[0], v.go.y & mstage[1], v.go.z & mstage[2], v.go.w & mstage[3]});
        else park_piece(tg, v.go);
        if constexpr (!ZEROS) {
            if (small_x) walk_park_guards(tx, gdx, v.xo, pad);   // (uniform)
            if (small_g) walk_park_guards(tg, gdg, v.go, pad);
        }
    };

    // ---- this thread's chunk ---------------------------------------------------------------------------------------------------
    const bool mine = tr < R && tr < Rn;
    const int b = b0 + tr;
    const WalkWindow<NA> wx = walk_window<ZEROS, NA, LEAN>(ji, small_x ? sx2 : d.cx2, S2, pad, slot0, mine, small_x);
    const WalkWindow<NA> wg = walk_window<ZEROS, NA, LEAN>(ji, (small_g ? sg2 : d.cg2) + L2, S2, pad, slot0, mine, small_g);
    const uint32_t row1 = static_cast<uint32_t>(RP) * 4u;
    const int px = (small_x ? sx2 : d.cx2) & 1, pg = ((small_g ? sg2 : d.cg2) + L2) & 1;   // half-word parity of the windows (uniform: rows are whole pieces)
    const float dP = static_cast<float>(d.dw[0]), dR = static_cast<float>(d.dw[1]), dC = static_cast<float>(d.dw[2]);
    const uint32_t my = mine ? static_cast<uint32_t>(b * S2 + ji) * 2u : kOOR;   // own chunk, bytes within a plane
    // CROP: the own chunk = window columns ji - L2 .. ji - L2 + 7 of window row b - L1, plane a - L0: the aligned pieces tc - 1 and tc
    const bool mine_g = mine && static_cast<unsigned>(b - L1) < static_cast<unsigned>(O1);
    const uint32_t myg = mine_g ? static_cast<uint32_t>((b - L1) * O2 + ji) * 2u : kOOR;
    const uint32_t myg_prev = (mine_g && tc > 0 && L2 > 0) ? myg - 16u : kOOR;
    struct OwnRaw { u4_t lo, hi; };
    auto load_own = [&](int a, bool have) {
        OwnRaw o;
        if constexpr (CROP) {
            const bool in = have && static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0);   // (uniform)
            const uint32_t so = in ? static_cast<uint32_t>(a - L0) * gplane_bytes : 0u;
            o.lo = __builtin_amdgcn_raw_buffer_load_b128(in ? gres : none, myg_prev, so, 0);
            o.hi = __builtin_amdgcn_raw_buffer_load_b128(in ? gres : none, myg, so, 0);
        } else {
            o.hi = __builtin_amdgcn_raw_buffer_load_b128(have ? gres : none, my, static_cast<uint32_t>(a) * plane_bytes, 0);
            o.lo = o.hi;
        }
        return o;
    };
    // the chunk as the wgrad sums pair it with the x dwords: CROP funnels L2 elements of the previous piece in front, and masks
    auto own_chunk = [&](const OwnRaw &o) {
        if constexpr (!CROP) {
            return o.hi;
        } else {
            u4_t r = o.hi;
            if (L2 == 1) {          // (uniform)
                r = u4_t{__builtin_amdgcn_alignbit(o.hi.x, o.lo.w, 16), __builtin_amdgcn_alignbit(o.hi.y, o.hi.x, 16),
                         __builtin_amdgcn_alignbit(o.hi.z, o.hi.y, 16), __builtin_amdgcn_alignbit(o.hi.w, o.hi.z, 16)};
            } else if (L2 == 2) {
                r = u4_t{o.lo.w, o.hi.x, o.hi.y, o.hi.z};
            }
            return u4_t{r.x & mown[0], r.y & mown[1], r.z & mown[2], r.w & mown[3]};
        }
    };
    auto lerp = [](float v1, float v2, float x) { return lerp1_fused<float>(v1, v2, x); };

    // row / column blend of one gradient plane at this thread's chunk (ACTIVE): B[e] = blend over rows b, b + 1 and columns
    // e, e + 1 of the staged plane.  PAR = half-word parity of the window.
    auto plane_blend = [&](auto par_tag, float (&B)[E]) {
        constexpr int PAR = decltype(par_tag)::value;
        uint32_t r0[5], r1[5];
        walk_read<PLAIN, NA, PAR>(tg, wg, 0u, r0);
        walk_read<PLAIN, NA, PAR>(tg, wg, row1, r1);
        float rb[E + 1];
#pragma unroll
        for (int k = 0; k <= E; ++k) {
            const int h = k + PAR;
            rb[k] = lerp(half_value<T>(r0[h >> 1], h & 1), half_value<T>(r1[h >> 1], h & 1), dR);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) B[e] = lerp(rb[e], rb[e + 1], dC);
    };
    // weight-gradient sums of one step: corners (plane pl, row hb) of x against the own gradient chunk, as the x dwords lie.
    // sa: the chunk as it lies against dwords PAR .. PAR + 3, sb: the chunk shifted by one element against dwords 0 .. 4.
    // Column offset 0 / 1 of the corner = (sa, sb) for an even window, (sb, sa) for an odd one.
    auto wgrad = [&](auto par_tag, const u4_t &g, const uint32_t (&x0)[2][5], uint32_t (&x1)[2][5], float (&sa)[2][2], float (&sb)[2][2]) {
        constexpr int PAR = decltype(par_tag)::value;
        walk_read<PLAIN, NA, PAR>(tx, wx, 0u, x1[0]);
        walk_read<PLAIN, NA, PAR>(tx, wx, row1, x1[1]);
        const uint32_t gq[4] = {g.x, g.y, g.z, g.w};
        const uint32_t gs[5] = {gq[0] << 16, __builtin_amdgcn_alignbit(gq[1], gq[0], 16), __builtin_amdgcn_alignbit(gq[2], gq[1], 16),
                                __builtin_amdgcn_alignbit(gq[3], gq[2], 16), gq[3] >> 16};
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const uint32_t(&w)[5] = pl ? x1[hb] : x0[hb];
#pragma unroll
                for (int i = 0; i < 4; ++i) sa[pl][hb] = dot2_packed<T>(gq[i], w[i + PAR], sa[pl][hb]);
#pragma unroll
                for (int i = 0; i < 5; ++i) sb[pl][hb] = dot2_packed<T>(gs[i], w[i], sb[pl][hb]);
            }
        }
    };
    using par0 = std::integral_constant<int, 0>;
    using par1 = std::integral_constant<int, 1>;

    // ---- the "+0" planes of the first step --------------------------------------------------------------------------------------
    uint32_t xa[2][5], xb[2][5];     // packed x windows of the two corner planes (alternating roles)
    float Ba[E], Bb[E];              // ACTIVE: blended gradient planes (alternating roles)
    float sa[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, sb[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // every load of the prologue is issued before the first wait: the first step's planes, the second's and the own chunk
    // travel together with the "+0" planes (one memory round trip per workgroup instead of two)
    constexpr int GA = ACTIVE ? 1 : 0;   // the gradient plane of step a: the "+1" corner plane / the plane the tap reads
    Staged stA, stB;
    OwnRaw gcur;
    {
        Staged v0;
        load_planes(xmap0(a0), ACTIVE ? gmap0(a0) : -1, v0);
        load_planes(xmap0(a0 + 1), gmap0(a0 + GA), stA);
        load_planes(a0 + 1 < a1 ? xmap0(a0 + 2) : -1, a0 + 1 < a1 ? gmap0(a0 + 1 + GA) : -1, stB);
        gcur = load_own(a0, true);
        walk_barrier();   // the tiles are zero
        park(v0);
        walk_barrier();
        if (px) {
            walk_read<PLAIN, NA, 1>(tx, wx, 0u, xa[0]);
            walk_read<PLAIN, NA, 1>(tx, wx, row1, xa[1]);
        } else {
            walk_read<PLAIN, NA, 0>(tx, wx, 0u, xa[0]);
            walk_read<PLAIN, NA, 0>(tx, wx, row1, xa[1]);
        }
        if constexpr (ACTIVE) {
            if (pg) plane_blend(par1{}, Ba);
            else plane_blend(par0{}, Ba);
        }
    }
    walk_barrier();   // the "+0" planes have been read

    auto flush = [&]() {   // fp32 sums of this wave -> its fp64 slots (fixed DPP tree, lane 63)
        const float v[8] = {sa[0][0], sa[1][0], sa[0][1], sa[1][1], sb[0][0], sb[1][0], sb[0][1], sb[1][1]};   // [kind][hb][pl]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = wave_total(v[i]);
            if ((tid & 63) == 63) wsum[wave * 8 + i] += static_cast<double>(t);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) sa[pl][hb] = sb[pl][hb] = 0.f;
    };

    // one step: `pend` holds the planes of step a and leaves with those of step a + 2 in flight; x0 / B0: the "+0" planes (in),
    // x1 / B1: the "+1" planes (out: the next step's "+0")
    auto walk_step = [&](int a, Staged &pend, const uint32_t (&x0)[2][5], uint32_t (&x1)[2][5], const float (&B0)[E], float (&B1)[E]) {
        park(pend);
        walk_barrier();
        const bool more = a + 2 < a1;
        load_planes(more ? xmap0(a + 3) : -1, more ? gmap0(a + 2 + GA) : -1, pend);
        {
            const u4_t gown = own_chunk(gcur);
            if (px) wgrad(par1{}, gown, x0, x1, sa, sb);
            else wgrad(par0{}, gown, x0, x1, sa, sb);
        }
        gcur = load_own(a + 1, a + 1 < a1);   // the next step's own chunk: in flight through the blends and the next staging
        u4_t res;
        if constexpr (ACTIVE) {
            if (pg) plane_blend(par1{}, B1);
            else plane_blend(par0{}, B1);
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = lerp(B0[e], B1[e], dP);
            Chunk<S, E> ch;
#pragma unroll
            for (int e = 0; e < E; ++e) ch.e[e] = narrow<T>(o[e]);
            __builtin_memcpy(&res, ch.e, 16);
        } else {   // the sparse shift: the window itself (bit patterns kept)
            uint32_t t[5];
            if (pg) {
                walk_read<PLAIN, NA, 1>(tg, wg, 0u, t);
                res = u4_t{__builtin_amdgcn_alignbit(t[1], t[0], 16), __builtin_amdgcn_alignbit(t[2], t[1], 16),
                           __builtin_amdgcn_alignbit(t[3], t[2], 16), __builtin_amdgcn_alignbit(t[4], t[3], 16)};
            } else {
                walk_read<PLAIN, NA, 0>(tg, wg, 0u, t);
                res = u4_t{t[0], t[1], t[2], t[3]};
            }
        }
        if constexpr (CROP) {   // grad_x is zero outside the window (mown: the row and the columns; the plane: uniform)
            const uint32_t keep = static_cast<unsigned>(a - L0) < static_cast<unsigned>(O0) ? ~0u : 0u;
            res = u4_t{res.x & mown[0] & keep, res.y & mown[1] & keep, res.z & mown[2] & keep, res.w & mown[3] & keep};
        }
        buffer_store_b128_soffset<kWalkStoreAux>(res, ores, my, static_cast<uint32_t>(a) * plane_bytes);
        if (((a - a0) & (kWalkFlush - 1)) == kWalkFlush - 1) flush();
        walk_barrier();   // everybody has read this step's planes
    };
    int a = a0;
    for (; a + 1 < a1; a += 2) {   // whole pairs: no condition between the steps (exact wait counts)
        walk_step(a, stA, xa, xb, Ba, Bb);
        walk_step(a + 1, stB, xb, xa, Bb, Ba);
    }
    if (a < a1) walk_step(a, stA, xa, xb, Ba, Bb);
    if (((a1 - a0) & (kWalkFlush - 1)) != 0) flush();
    walk_barrier();
    // ---- the workgroup's record: per-corner sums -> the corner-difference sums (corner_diffs is linear) ------------------------
    if (tid < 8) {
        double s[8];   // [kind][hb][pl]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s[i] = 0.0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) s[i] += wsum[w * 8 + i];
        }
        // corner q = pl | hb << 1 | col << 2 (step_backward's order); col = kind for an even x window, 1 - kind for an odd one
        double v[8], df[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pl = q & 1, hb = (q >> 1) & 1, col = q >> 2;
            const int kind_even = col, kind_odd = 1 - col;
            const double se = s[kind_even * 4 + hb * 2 + pl], so = s[kind_odd * 4 + hb * 2 + pl];
            v[q] = px ? so : se;
        }
        corner_diffs<3, double>(v, df);
        double out = df[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) out = tid == i ? df[i] : out;
        p.partials[static_cast<size_t>(bid) * 8 + tid] = out;
    }
}

template <typename T, bool ACTIVE, bool ZEROS, bool CROP = false>
__global__ __launch_bounds__(kThreads) void walk_backward16(const StepParams p) {
    if constexpr (ZEROS) {
        walk_backward16_body<T, ACTIVE, true, false, CROP>(p);
    } else {
        const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
        if (bid >= p.total_steps) return;
        const uint32_t plane = fdiv(bid, p.d_spp);
        const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
        const int cx2 = p.desc[c].cx2, cg2 = p.desc[c].cg2;
        const int per2 = map_period(p.S2, p.pad);
        const int sx2 = (per2 && 2 * cx2 > per2) ? cx2 - per2 : cx2, sg2 = (per2 && 2 * cg2 > per2) ? cg2 - per2 : cg2;
        const bool lean = p.cpr >= 2 && sx2 >= -kWalkSmall && sx2 <= kWalkSmall && sg2 >= -kWalkSmall && sg2 <= kWalkSmall;   // (uniform)
        if (lean) walk_backward16_body<T, ACTIVE, false, true>(p);
        else walk_backward16_body<T, ACTIVE, false, false>(p);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// walk_forward16: the 3-D interpolating forward of 16-bit tensors, the same walk with the same tile layout: one plane of x
// staged per step, its row / column blend computed once and carried to the next step as the "+0" plane (8 fp32 values).
// Reference: kernels/shifts_kernels.h:156-220 (:187-205), interpolation.h:34-40; weights cuda/shifts_cuda.cu:168-183.
// ---------------------------------------------------------------------------------------------------------------------
// CROP (round 6, zeros padding): the output is a WINDOW [O0, O1, O2] that begins at (L0, L1, L2) of the shifted volume
// (ops/shifts.cpp:93-135).  A workgroup owns R rows of the window and walks through its O0 planes; the staged source rows and planes
// are offset by (L1, L0), the crop along the row is one more column shift of the window (cs2 - L2).  Thread (tr, tc) owns window
// columns 8 tc ..: window rows of an even number of elements start on a dword, the row's last piece leaves as 1 - 4 dwords.
// ACTIVE = false (CROP only): the sparse shift of a cropped volume -- the modules' default mode -- on the same skeleton: one staged
// plane per step, the window itself leaves (bit patterns kept), nothing is carried (weights: cuda/shifts_cuda.cu:168-183, round half even).
template <typename T, bool ZEROS, bool CROP = false, bool ACTIVE = true>
__global__ __launch_bounds__(kThreads) void walk_forward16(const FwdParams p) {
    using S = typename T::S;
    static_assert(sizeof(S) == 2, "16-bit element types");
    static_assert(!CROP || ZEROS, "the cropped walk: zeros padding");
    static_assert(ACTIVE || CROP, "the sparse form: cropped volumes (the uncropped ones have their one-step kernels)");
    constexpr int E = 8;
    constexpr int NA = ZEROS ? 5 : 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const tx = smem;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const int pad = ZEROS ? 0 : p.pad;
    float wv[3];
    load_weights_nd<float>(p.w, p.wkind, c, 3, wv);
    float rr[3], dn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {   // weights_init_forward, active: floor + fraction; sparse: round half even
        rr[d] = ACTIVE ? floorf(wv[d]) : rintf(wv[d]);
        dn[d] = ACTIVE ? wv[d] - rr[d] : 0.f;
    }
    const int cs0 = __builtin_amdgcn_readfirstlane(canon_shift(static_cast<int64_t>(rr[0]), p.S0, pad, p.d_per0));
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_shift(static_cast<int64_t>(rr[1]), p.S1, pad, p.d_per1));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_shift(static_cast<int64_t>(rr[2]), p.S2, pad, p.d_per2));
    const float dP = dn[0], dR = dn[1], dC = dn[2];

    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr;
    const int O0 = CROP ? p.O0 : S0, O1 = CROP ? p.O1 : S1, O2 = CROP ? p.O2 : S2;
    const int L0 = CROP ? p.L0 : 0, L1 = CROP ? p.L1 : 0, L2 = CROP ? p.L2 : 0;
    const int b0 = step * R;
    const int Rn = min(R, O1 - b0);
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    const int ji = tc * E;
    const int RP = cpr + kWalkGuard;
    for (int o = tid * 16; o < kWalkTileBytes; o += kThreads * 16)
        *reinterpret_cast<u4_t *>(__builtin_assume_aligned(smem + o, 16)) = u4_t{0u, 0u, 0u, 0u};

    const bool own = tr <= R && tr <= Rn;
    const int sx_own = own ? row_map(b0 + tr + L1, cs1, S1, pad) : -1;
    constexpr uint32_t kOOR = 0x80000000u;
    constexpr int kRsrcFlags = 0x00020000;
    const uint32_t plane_bytes = static_cast<uint32_t>(S1) * static_cast<uint32_t>(S2) * 2u;
    const uint32_t vol_bytes = static_cast<uint32_t>(S0) * plane_bytes;
    const uint32_t oplane_bytes = CROP ? static_cast<uint32_t>(O1) * static_cast<uint32_t>(O2) * 2u : plane_bytes;
    const char *xp = reinterpret_cast<const char *>(static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane);
    char *op = reinterpret_cast<char *>(static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, vol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(op, 0, CROP ? static_cast<uint32_t>(O0) * oplane_bytes : vol_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xp), 0, 0, kRsrcFlags);
    const uint32_t vx_own = sx_own >= 0 ? static_cast<uint32_t>(sx_own * S2 + ji) * 2u : kOOR;
    const uint32_t park_at = static_cast<uint32_t>(own ? kWalkMargin + tr * RP + tc : kWalkDump0 + (tid & 63)) * 4u;
    auto load_plane = [&](int pa) {   // source plane (uniform; -1: fill)
        return __builtin_amdgcn_raw_buffer_load_b128(pa >= 0 ? xres : none, vx_own, pa >= 0 ? static_cast<uint32_t>(pa) * plane_bytes : 0u, 0);
    };
    const bool mine = tr < R && tr < Rn && ji < O2;   // (CROP: the window's rows hold fewer pieces than the source's)
    const int slot0 = kWalkMargin + tr * RP;
    const int per2 = map_period(S2, pad);
    const int ss2 = (!ZEROS && per2 && 2 * cs2 > per2) ? cs2 - per2 : cs2;   // the signed column shift
    const bool small_x = !ZEROS && cpr >= 2 && ss2 >= -kWalkSmall && ss2 <= kWalkSmall;
    const WalkGuards gdx = walk_guards(small_x, own, tc == 0, tc == cpr - 1, pad, slot0, cpr, tid & 63);
    auto park = [&](const u4_t &v) {
        uint32_t *q = reinterpret_cast<uint32_t *>(tx + park_at);
        q[0] = v.x;
        q[kWalkSlots] = v.y;
        q[2 * kWalkSlots] = v.z;
        q[3 * kWalkSlots] = v.w;
        if constexpr (!ZEROS) {
            if (small_x) walk_park_guards(tx, gdx, v, pad);   // (uniform: the folded row ends, see WalkWindow)
        }
    };
    const WalkWindow<NA> wx = walk_window<ZEROS, NA>(ji, (small_x ? ss2 : cs2) - L2, S2, pad, slot0, mine, small_x);
    const uint32_t row1 = static_cast<uint32_t>(RP) * 4u;
    const int px = ((small_x ? ss2 : cs2) - L2) & 1;
    const uint32_t my = mine ? static_cast<uint32_t>((b0 + tr) * O2 + ji) * 2u : kOOR;
    const int ndw = CROP ? min(4, (O2 - ji) >> 1) : 4;   // dwords of the chunk inside the row (the window's last piece: 1 - 4)
    auto lerp = [](float v1, float v2, float x) { return lerp1_fused<float>(v1, v2, x); };
    auto plane_blend = [&](auto par_tag, float (&B)[E]) {
        constexpr int PAR = decltype(par_tag)::value;
        uint32_t r0[5], r1[5];
        walk_read<ZEROS, NA, PAR>(tx, wx, 0u, r0);
        walk_read<ZEROS, NA, PAR>(tx, wx, row1, r1);
        float rb[E + 1];
#pragma unroll
        for (int k = 0; k <= E; ++k) {
            const int h = k + PAR;
            rb[k] = lerp(half_value<T>(r0[h >> 1], h & 1), half_value<T>(r1[h >> 1], h & 1), dR);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) B[e] = lerp(rb[e], rb[e + 1], dC);
    };
    using par0 = std::integral_constant<int, 0>;
    using par1 = std::integral_constant<int, 1>;

    float Ba[E], Bb[E];
    u4_t stA, stB;
    if constexpr (ACTIVE) {
        const u4_t v0 = load_plane(row_map(L0, cs0, S0, pad));
        stA = load_plane(row_map(L0 + 1, cs0, S0, pad));
        stB = load_plane(1 < O0 ? row_map(L0 + 2, cs0, S0, pad) : -1);
        walk_barrier();   // the tile is zero
        park(v0);
        walk_barrier();
        if (px) plane_blend(par1{}, Ba);
        else plane_blend(par0{}, Ba);
    } else {   // the sparse shift: step a reads the one plane under output plane a
        stA = load_plane(row_map(L0, cs0, S0, pad));
        stB = load_plane(1 < O0 ? row_map(L0 + 1, cs0, S0, pad) : -1);
    }
    walk_barrier();
    auto walk_step = [&](int a, u4_t &pend, const float (&B0)[E], float (&B1)[E]) {
        park(pend);
        walk_barrier();
        u4_t res;
        if constexpr (ACTIVE) {
            pend = load_plane(a + 2 < O0 ? row_map(a + 3 + L0, cs0, S0, pad) : -1);
            if (px) plane_blend(par1{}, B1);
            else plane_blend(par0{}, B1);
            Chunk<S, E> ch;
#pragma unroll
            for (int e = 0; e < E; ++e) ch.e[e] = narrow<T>(lerp(B0[e], B1[e], dP));
            __builtin_memcpy(&res, ch.e, 16);
        } else {
            pend = load_plane(a + 2 < O0 ? row_map(a + 2 + L0, cs0, S0, pad) : -1);
            uint32_t t[5];
            if (px) {
                walk_read<ZEROS, NA, 1>(tx, wx, 0u, t);
                res = u4_t{__builtin_amdgcn_alignbit(t[1], t[0], 16), __builtin_amdgcn_alignbit(t[2], t[1], 16),
                           __builtin_amdgcn_alignbit(t[3], t[2], 16), __builtin_amdgcn_alignbit(t[4], t[3], 16)};
            } else {
                walk_read<ZEROS, NA, 0>(tx, wx, 0u, t);
                res = u4_t{t[0], t[1], t[2], t[3]};
            }
        }
        if constexpr (CROP) {
            const uint32_t so = static_cast<uint32_t>(a) * oplane_bytes;
            if (ndw == 4) {
                buffer_store_b128_soffset<kWalkStoreAux>(res, ores, my, so);
            } else if (mine) {   // the row's last piece: 1 - 3 dwords (the next row begins behind them)
                typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
                if (ndw & 2) __builtin_amdgcn_raw_buffer_store_b64(u2_t{res.x, res.y}, ores, my, so, 0);
                if (ndw & 1) __builtin_amdgcn_raw_buffer_store_b32((ndw & 2) ? res.z : res.x, ores, my + ((ndw & 2) ? 8u : 0u), so, 0);
            }
        } else {
            buffer_store_b128_soffset<kWalkStoreAux>(res, ores, my, static_cast<uint32_t>(a) * plane_bytes);
        }
        walk_barrier();
    };
    int a = 0;
    for (; a + 1 < O0; a += 2) {
        walk_step(a, stA, Ba, Bb);
        walk_step(a + 1, stB, Bb, Ba);
    }
    if (a < O0) walk_step(a, stA, Ba, Bb);
}

struct WalkPlan {
    int cpr, R, spp;
    uint64_t total;
    size_t off_desc, bytes;
};

// rows: the rows the workgroups of an (n, c) volume share -- S1, or the window's O1 for the cropped forward
WalkPlan walk_plan(const Geometry &g, int64_t rows = 0) {
    WalkPlan w{};
    if (rows <= 0) rows = g.S[1];
    w.cpr = static_cast<int>(g.S[2] * 2 / 16);
    if (w.cpr < 1) w.cpr = 1;
    // (R + 1) * cpr <= 256: every staged piece has its thread; the tile: margin + (R + 1) * (cpr + guard) <= 448 slots
    int rmax = kThreads / w.cpr - 1;
    rmax = std::min(rmax, (kWalkDump0 - kWalkMargin) / (w.cpr + kWalkGuard) - 1);
    rmax = std::max(1, std::min<int>(rmax, static_cast<int>(rows)));
    w.spp = static_cast<int>((rows + rmax - 1) / rmax);
    w.R = static_cast<int>((rows + w.spp - 1) / w.spp);   // balanced steps: 112 rows of 14 pieces -> 7 steps of 16 rows
    w.total = static_cast<uint64_t>(g.N) * g.C * w.spp;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    w.off_desc = up(w.total * 8 * sizeof(double));
    w.bytes = w.off_desc + up(static_cast<size_t>(g.C) * sizeof(ChanDesc));
    return w;
}

bool walk16_volume_ok(const Geometry &g, int dtype) {
    if ((dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) || g.nd != 3 || g.K[0] > 0 || g.S[0] < 2) return false;
    if (g.S[1] < 1 || (g.S[2] * 2) % 16 != 0 || g.S[2] * 2 / 16 > kThreads / 2 || g.S[2] > 32000) return false;
    if (g.S[0] * g.S[1] * g.S[2] * 2 >= (1LL << 31)) return false;   // (one buffer resource spans an (n, c) volume)
    return walk_plan(g).total + 8 < (1ull << 31);
}

bool walk16_cropped(const Geometry &g) {
    for (int d = 0; d < 3; ++d)
        if (g.O[d] != g.S[d] || g.L[d] != 0) return true;
    return false;
}

bool walk16_geometry_ok(const Geometry &g, int dtype) { return walk16_volume_ok(g, dtype) && !walk16_cropped(g); }

// walk_backward16<.., CROP>: zeros padding, a window with rows of an even number of elements (every window row starts on a dword)
// that begins at most two columns into the volume's rows (the own chunk's funnel), every window dim at least 2 (a size-1 dim ignores
// its shift: the general kernels).  Knob 35 bit 11 keeps crop_backward3.
bool walk16_crop_geometry_ok(const Geometry &g, int dtype) {
    return walk16_volume_ok(g, dtype) && walk_crop_window_ok(g) && g.O[2] % 2 == 0;
}

}  // namespace

// the 3-D interpolating forward of fp16 / bf16 tensors: contiguous, no crop, rows of whole 16-byte pieces, float weights of the
// tensor's dtype
bool walk16_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[2] == 1 || (g_step_tune[3] & 16)) return false;   // knob 34 = 1: no forwards through LDS; knob 35 bit 4: no walk kernels
    // (a window: walk_forward16<.., CROP> -- zeros padding, window rows of an even number of elements, every window dim at least 2)
    bool crop_ok = walk16_volume_ok(g, dtype) && walk16_cropped(g) && g.pad == 0 && !(g_step_tune[3] & 2048) && g.O[2] % 2 == 0;
    for (int d = 0; d < 3; ++d) crop_ok = crop_ok && g.O[d] >= 2 && g.L[d] >= 0 && g.L[d] + g.O[d] <= g.S[d];
    if (!(g.active ? (walk16_geometry_ok(g, dtype) || crop_ok) : crop_ok)) return false;   // (the sparse shift: cropped volumes only)
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    return reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
}

int walk16_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const bool crop = walk16_cropped(g);
    const WalkPlan W = walk_plan(g, crop ? g.O[1] : 0);
    FwdParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.nd = 3;
    p.pad = g.pad;
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
    p.cpr = p.xppr = W.cpr;
    p.R = W.R;
    p.spp = p.spv = W.spp;
    p.total_steps = static_cast<uint32_t>(W.total);
    p.steps_per_xcd = static_cast<uint32_t>((W.total + 7) / 8);
    p.d_spp = p.d_spv = make_fastdiv(static_cast<uint32_t>(W.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = p.d_xppr = make_fastdiv(static_cast<uint32_t>(W.cpr));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    note_kernel(crop ? (g.active ? "walk_forward16_crop" : "walk_forward16_crop_sparse") : "walk_forward16");
    const bool zeros = g.pad == 0;
    if (crop && !g.active) {   // (a raw copy of the window: one instantiation for both 16-bit types)
        hipLaunchKernelGGL((walk_forward16<f16_t, true, true, false>), grid, block, kWalkTileBytes, st, p);
    } else if (crop) {
        if (dtype == SHIFTND_F16) hipLaunchKernelGGL((walk_forward16<f16_t, true, true>), grid, block, kWalkTileBytes, st, p);
        else hipLaunchKernelGGL((walk_forward16<bf16_t, true, true>), grid, block, kWalkTileBytes, st, p);
    } else if (dtype == SHIFTND_F16) {
        if (zeros) hipLaunchKernelGGL((walk_forward16<f16_t, true>), grid, block, kWalkTileBytes, st, p);
        else hipLaunchKernelGGL((walk_forward16<f16_t, false>), grid, block, kWalkTileBytes, st, p);
    } else {
        if (zeros) hipLaunchKernelGGL((walk_forward16<bf16_t, true>), grid, block, kWalkTileBytes, st, p);
        else hipLaunchKernelGGL((walk_forward16<bf16_t, false>), grid, block, kWalkTileBytes, st, p);
    }
    return SHIFTND_OK;
}

// the 3-D backward of fp16 / bf16 tensors (both shifts, every padding): contiguous, no crop, rows of whole 16-byte pieces
bool walk16_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g_step_tune[0] == 1 || (g_step_tune[3] & 16)) return false;   // knob 32 = 1: never; knob 35 bit 4: no walk kernels
    if (!walk16_geometry_ok(g, dtype) && !walk16_crop_geometry_ok(g, dtype)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O) || !dense(g.gs, g.N, g.C, g.S)) return false;
    return reinterpret_cast<uintptr_t>(go) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(gx) % 16 == 0;
}

size_t walk16_backward_workspace(const Geometry &g, int dtype) { return walk16_volume_ok(g, dtype) ? walk_plan(g).bytes : 0; }

int walk16_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                    hipStream_t st) {
    const WalkPlan W = walk_plan(g);
    StepParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    char *ws = static_cast<char *>(workspace);
    p.partials = reinterpret_cast<double *>(ws);
    p.desc = reinterpret_cast<ChanDesc *>(ws + W.off_desc);
    p.colx = p.colg = nullptr;   // (no column tables: the kernel folds its maps itself)
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.wkind = dtype;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.pad = g.pad;
    p.nd = 3;
    p.S0 = static_cast<int>(g.S[0]);
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.cpr = W.cpr;
    p.R = W.R;
    p.spp = p.spv = W.spp;
    p.walk_planes = p.S0;
    p.total_steps = static_cast<uint32_t>(W.total);
    p.steps_per_xcd = static_cast<uint32_t>((W.total + 7) / 8);
    p.d_spp = p.d_spv = make_fastdiv(static_cast<uint32_t>(W.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(W.cpr));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const size_t lds = 2 * kWalkTileBytes + (kThreads / 64) * 8 * sizeof(double);
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    const bool active = g.active != 0, zeros = g.pad == 0;
    const bool crop = walk16_cropped(g);
    if (crop) {   // the window: sizes in wO0 / wO1 / wO2, its first plane / row / column in wL0 / wL1 / wL2 (walk_backward16<.., CROP>)
        p.wO0 = static_cast<int>(g.O[0]);
        p.wO1 = static_cast<int>(g.O[1]);
        p.wO2 = static_cast<int>(g.O[2]);
        p.wL0 = static_cast<int>(g.L[0]);
        p.wL1 = static_cast<int>(g.L[1]);
        p.wL2 = static_cast<int>(g.L[2]);
        p.g_plane = g.O[0] * g.O[1] * g.O[2];
        p.crop = 1;
    }
    note_kernel(crop ? (active ? "walk_backward16_crop" : "walk_backward16_crop_sparse") : (active ? "walk_backward16" : "walk_backward16_sparse"));
#define SHIFTND_WALK16(TT) \
    { \
        launch_step_prep(TT::kDtype, active, p, st); \
        if (crop) { \
            if (active) hipLaunchKernelGGL((walk_backward16<TT, true, true, true>), grid, block, lds, st, p); \
            else hipLaunchKernelGGL((walk_backward16<TT, false, true, true>), grid, block, lds, st, p); \
        } else if (active) { \
            if (zeros) hipLaunchKernelGGL((walk_backward16<TT, true, true>), grid, block, lds, st, p); \
            else hipLaunchKernelGGL((walk_backward16<TT, true, false>), grid, block, lds, st, p); \
        } else { \
            if (zeros) hipLaunchKernelGGL((walk_backward16<TT, false, true>), grid, block, lds, st, p); \
            else hipLaunchKernelGGL((walk_backward16<TT, false, false>), grid, block, lds, st, p); \
        } \
        launch_step_reduce(TT::kDtype, 3, p, gw, st); \
    }
    if (dtype == SHIFTND_F16) SHIFTND_WALK16(f16_t) else SHIFTND_WALK16(bf16_t)
#undef SHIFTND_WALK16
    return SHIFTND_OK;
}

}  // namespace shiftnd
