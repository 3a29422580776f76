// shiftnd_step.hip -- one-step workgroups: the backward pass and the forwards of contiguous 2-D (and 3-D) problems as a
// linear sweep of short workgroups (gfx950 / MI355X).  DESIGN section 3.16.
//
// What the memory system of the MI355X rewards (tools/stream_probe --mock): many SHORT workgroups
// that are dispatched in address order, each moving a few KB and exiting -- a plain 2-read-1-write stream staged
// through LDS reaches 6.5 TB/s in that shape, against 5.4 TB/s when a workgroup walks a 50 KB band of its own (the shape
// of plane_backward_lds) and 5.2 TB/s for a whole plane per workgroup: the fewer independent sweep fronts the DRAM
// sees, the better (8 fronts, one per XCD, are as good as one).  So here a workgroup owns ONE step -- U * R consecutive
// rows of one (n, c) plane, R = 256 / (chunks per row), U = 1 or 2 row groups per thread -- and the grid is every step of
// the tensor in memory order, the workgroups of an XCD (blockIdx % 8) owning a contiguous eighth of it.
//
// A workgroup that lives for one step cannot build per-channel index maps in LDS, nor amortise a long scalar prologue
// (all waves of a workgroup run the scalar part, one scalar unit per CU): the padding mode is a template parameter, row
// sources are folded arithmetically (fold_index: 2 - 5 VALU), per-channel state comes through the scalar cache.
//
//   step_prep / step_backward / step_reduce   backward pass (2-D by default, 3-D on request; fused average-pool tail for
//                          the 2-D sparse shift).  step_prep (one workgroup per channel) writes ChanDesc -- canonical
//                          shifts, scatter shift, fractions -- and, for the paddings other than zeros, the column state of
//                          every chunk; step_backward stages the R + 1 corner rows of x and the R rows of grad_out
//                          (interpolating: + the rows grad_x blends) by LDS-DMA, nontemporal, one barrier, the arithmetic of
//                          plane_backward_lds, one store per thread; the sparse shift runs in scatter form.  Weight
//                          gradient: per-thread sums of g * corner difference, a fixed DPP tree per wave, the waves added
//                          in fp64 -> partials[step][NDIFF]; step_reduce adds a channel's steps in a fixed order and applies
//                          the per-channel blends once.  Deterministic, no atomics.
//   step_gather_forward         sparse-shift / quantized forward of 4- / 8-byte elements: one element-aligned 16-byte load, one
//                          store, no LDS.
//   step_gather_forward_small   ... of 1- / 2-byte elements: two aligned 16-byte loads and a workgroup-uniform byte funnel.
//   step_forward_lds            forwards through LDS: interpolating (2-D every float dtype, 3-D 4- / 8-byte) and the 2-byte
//                          sparse shift; 3-D blends shared along the reference's nesting.
//   step_active_forward_direct  interpolating forward by raw-buffer windows, no LDS (on request: slower than the LDS form).
//   walk_forward / walk_backward  3-D problems as a walk through the PLANES (the one exception to "one step per workgroup"): a
//                          workgroup owns R rows of one (n, c) volume and steps through its planes; the "+1" corner plane of a
//                          step is the "+0" plane of the next and stays in registers, so a step stages one plane per tensor
//                          and reads two windows instead of four (the one-step 3-D kernels are bound by exactly that work).
//                          Forward: interpolating; POOL: the module's average pool as the epilogue.  Backward: interpolating
//                          and sparse shift; packed v_dot2c corner sums for 16-bit data, register-prefetched staging,
//                          conflict-free window reads (lds_window6); POOL: the pooled gradient expanded on its way into LDS.
//                          Partials and reduction as for step_backward (one record per workgroup).
//
// Reference behaviour restated: kernels/shifts_kernels.h:156-220 (forward), :222-327 (backward), :132-154 (weight
// gradients), :532-571 (quantized); kernels/interpolation.h:3-61; cuda/shifts_cuda.cu:168-199, :202-345 (weight
// preparation, launch).  Roofline: HBM; forward 2 s bytes per element, backward 3 s.
#include "shiftnd_step.hpp"

namespace shiftnd {

thread_local int g_step_tune[5] = {0, 0, 0, 0, 0};   // knobs 32..36 / 38: see shiftnd_step.hpp

namespace {

// U = row groups per thread: a workgroup owns U * R rows.  U = 2 halves the per-workgroup scalar work and the column-state
// prologue per byte -- for the variants that are bound by instruction issue rather than by memory order (16-bit data, the
// interpolating shift); the light fp32 sparse shift keeps U = 1 (the tighter sweep front: DESIGN 3.16).
template <typename T, int ND, bool ACTIVE, int PAD, bool POOL = false, int U = 1>
__global__ __launch_bounds__(kThreads) void step_backward(const StepParams p) {
    static_assert(!POOL || ND == 2, "the fused pool tail is 2-D here");
    static_assert(!POOL || U == 1, "the pooled variant keeps one row group per thread");
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int REC = RecSize<E>::N;
    constexpr bool SCAT = !ACTIVE && ND == 2;  // 3-D sparse shift: gather form (one staged row per grad_x row)
    constexpr int NDIFF = WDiff<ND>::N;
    constexpr int NP = ND == 3 ? 2 : 1;        // planes a step's corners come from
    constexpr int NCC = 1 << (ND - 1);         // corner rows per element
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;  // 64-byte pad: see lds_read_row

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);  // XCD-contiguous step ids
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spv);          // (n, c)
    const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spv);
    const int a = ND == 3 ? static_cast<int>(fdiv(vstep, p.d_spp)) : 0;
    const int step = static_cast<int>(vstep) - a * p.spp;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int R = p.R, S0 = p.S0, S1 = p.S1, S2 = p.S2, cpr = p.cpr;
    const int RT = U * R;  // rows of this step
    const int b0 = step * RT;
    const int Rn = min(RT, S1 - b0);
    // staged groups, in rows of the tile: X planes [NP][R + 1], G [R], GS: active [NP][R + 1], 3-D sparse [R], 2-D sparse none
    const int NX = NP * (RT + 1), NG = RT;
    const int NGS = ACTIVE ? NP * (RT + 1) : (SCAT ? 0 : RT);
    const int npieces = (NX + NG + NGS) * cpr;
    const int RB = S2 * static_cast<int>(sizeof(S));
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * (POOL ? p.g_plane : p.x_plane);
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane + static_cast<int64_t>(a) * S1 * S2;
    // source planes of the corners (3-D; uniform): -1 = fill
    int pax[NP], pag[NP];
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        pax[h] = ND == 3 ? row_map_t<PAD>(a + h, d.cx0, S0) : 0;
        pag[h] = ND == 3 ? row_map_t<PAD>(a + h, d.cg0, S0) : 0;
    }

    // ---- the thread's chunk: column state through both maps -----------------------------------------------------
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    const int ji = tc * E;
    ColState<E> xm, gm;
    if constexpr (PAD == 0) {  // zeros: column j0 + e reads column j0 + e - shift when that is a column
        auto affine_state = [&](int cs) {
            ColState<E> st;
            st.base = ji - cs;
            if (st.base + E < 0 || st.base >= S2) st.base = 0;  // no column of the chunk has a source: every entry is -1 below
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) st.cm[e] = (ji - cs + e >= 0 && ji - cs + e < S2) ? ji - cs + e : -1;
            return st;
        };
        xm = affine_state(d.cx2);
        gm = affine_state(d.cg2);
    } else {  // one vector load per map, consumed behind the barrier
        const size_t rec = (static_cast<size_t>(c) * cpr + tc) * REC;
        xm = load_colstate<E>(p.colx + rec);
        gm = load_colstate<E>(p.colg + rec);
    }

    // ---- stage the rows: aligned 16-byte pieces.  Thread (tr, tc) moves piece tc of row tr of every group (its own
    // chunk position: no index arithmetic), the first cpr threads move the extra corner row of the groups that have one;
    // a wave's pieces are consecutive, the LDS destination is a wave-uniform base (the hardware adds lane * 16).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma = [&](const S *base, int src_row, int col_piece, int lds_piece0) {
        // uniform base + 32-bit lane offset (planes are < 2^30 elements): the SGPR-base address form
        const uint32_t off = static_cast<uint32_t>(src_row * S2 + col_piece * E) * static_cast<uint32_t>(sizeof(S));
        char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(base) + off),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 2 /* nt */);
    };
    // POOL: a row of the unpooled gradient is its pooled row expanded: g(b, j) = grad_pooled[b / K1][j / K2] / (window size),
    // rounded to the storage type like the two-step sequence (ATen's avg_pool backward); a thread expands its own piece
    // into the tile (the loads of all its pieces first, so that their latencies overlap each other and the row DMA)
    // (three NAMED pieces per thread -- the step's own row, the row grad_x blends, the + 1 corner row -- each a value: arrays of
    //  chunks indexed in a loop, filled through references, stayed in scratch: 64 bytes per lane in round 3)
    struct Pooled {
        Chunk<S, E> raw;   // the pooled elements under the piece (E / 2 of them when K2 == 2)
        int cnt;           // rows of the pooled row's window
        int dst;           // tile piece (-1: none)
        int col;
    };
    auto pooled_load = [&](int grow, int col_piece, int dst) {
        Pooled q;
        q.dst = dst;
        q.col = col_piece;
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(grow), p.d_k1));
        q.cnt = min(p.K1, S1 - pr * p.K1);
        const S *prow = gp + static_cast<int64_t>(pr) * p.P2;
        S zero;
        __builtin_memset(&zero, 0, sizeof(S));
#pragma unroll
        for (int e = 0; e < E; ++e) q.raw.e[e] = zero;
        if (p.K2 == 2 && E % 2 == 0) {
            const Chunk<S, (E >= 2 ? E / 2 : 1)> h = load_chunk<S, (E >= 2 ? E / 2 : 1)>(prow + col_piece * (E / 2));
#pragma unroll
            for (int e = 0; e < E / 2; ++e) q.raw.e[e] = h.e[e];
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) q.raw.e[e] = prow[fdiv(static_cast<uint32_t>(col_piece * E + e), p.d_k2)];
        }
        return q;
    };
    auto pooled_store = [&](const Pooled &q) {
        Chunk<S, E> out;
        if (p.K2 == 2 && E % 2 == 0) {
#pragma unroll
            for (int h = 0; h < E / 2; ++h) {
                const S v = narrow<T>(div_count<CT>(widen<T>(q.raw.e[h]), q.cnt * 2));
                out.e[2 * h] = v;
                out.e[2 * h + 1] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pc = static_cast<int>(fdiv(static_cast<uint32_t>(q.col * E + e), p.d_k2));
                out.e[e] = narrow<T>(div_count<CT>(widen<T>(q.raw.e[e]), q.cnt * min(p.K2, S2 - pc * p.K2)));
            }
        }
        __builtin_memcpy(__builtin_assume_aligned(tile + q.dst * 16, 16), out.e, 16);
    };
    Pooled pqA, pqB, pqC;
    pqA.dst = pqB.dst = pqC.dst = -1;
    pqA.cnt = pqB.cnt = pqC.cnt = 1;
    pqA.col = pqB.col = pqC.col = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
    if (tr < R) {
        const int vtr = tr + u * R, vtid = tid + u * R * cpr;  // this row group's row / piece index
        const int sx = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cx1, S1) : -1;  // corner rows of x: m1[b0 + tr]
#pragma unroll
        for (int h = 0; h < NP; ++h)
            if (sx >= 0 && pax[h] >= 0) dma(xp, pax[h] * S1 + sx, tc, h * (RT + 1) * cpr + u * R * cpr);
        if (vtr < Rn) {  // the incoming gradient at the rows themselves
            if constexpr (POOL) pqA = pooled_load(b0 + vtr, tc, NX * cpr + vtid);
            else dma(gp, a * S1 + b0 + vtr, tc, NX * cpr + u * R * cpr);
        }
        if constexpr (ACTIVE) {
            const int sg = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cg1, S1) : -1;  // the rows grad_x blends: g1[b0 + tr]
            if constexpr (POOL) {
                if (sg >= 0) pqB = pooled_load(sg, tc, (NX + NG) * cpr + vtid);
            } else {
#pragma unroll
                for (int h = 0; h < NP; ++h)
                    if (sg >= 0 && pag[h] >= 0) dma(gp, pag[h] * S1 + sg, tc, (NX + NG + h * (RT + 1)) * cpr + u * R * cpr);
            }
        } else if constexpr (!SCAT) {
            const int sg = vtr < Rn ? row_map_t<PAD>(b0 + vtr, d.cg1, S1) : -1;  // 3-D sparse shift: the one row grad_x copies
            if (sg >= 0 && pag[0] >= 0) dma(gp, pag[0] * S1 + sg, tc, (NX + NG) * cpr + u * R * cpr);
        }
    }
    }
    if (Rn == RT && tid < cpr) {  // the + 1 corner row of a full step (a ragged last step has it among its first R rows)
        const int sx = row_map_t<PAD>(b0 + RT, d.cx1, S1);
#pragma unroll
        for (int h = 0; h < NP; ++h)
            if (sx >= 0 && pax[h] >= 0) dma(xp, pax[h] * S1 + sx, tid, (h * (RT + 1) + RT) * cpr);
        if constexpr (ACTIVE) {
            const int sg = row_map_t<PAD>(b0 + RT, d.cg1, S1);
            if constexpr (POOL) {
                if (sg >= 0) pqC = pooled_load(sg, tid, (NX + NG + RT) * cpr + tid);
            } else {
#pragma unroll
                for (int h = 0; h < NP; ++h)
                    if (sg >= 0 && pag[h] >= 0) dma(gp, pag[h] * S1 + sg, tid, (NX + NG + h * (RT + 1) + RT) * cpr);
            }
        }
    }
    if constexpr (POOL) {   // (every load above is in flight before the first conversion)
        if (pqA.dst >= 0) pooled_store(pqA);
        if constexpr (ACTIVE) {
            if (pqB.dst >= 0) pooled_store(pqB);
            if (pqC.dst >= 0) pooled_store(pqC);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    CT part[NDIFF];
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) part[i] = CT(0);
    // 16-bit data, 2-D sparse shift (C5): the windows stay packed.  grad_x is a raw copy of the gradient window; the
    // weight-gradient sums are v_dot2c products of packed pairs of x and of the incoming gradient -- four per-corner sums per
    // thread, differenced once per step -- instead of unpack, widen, subtract, multiply-add per element (walk_backward: -30 % of
    // the kernel's time); windows are read as two aligned ds_read_b128 (lds_window6: no bank conflicts) and masked per dword.
#ifndef SHIFTND_STEP_PK
#define SHIFTND_STEP_PK 1
#endif
    constexpr bool PK = SHIFTND_STEP_PK && sizeof(S) == 2 && SCAT;
    // ... and the 2-D interpolating shift of 16-bit data shares the packed corner sums (its grad_x blends keep their widened
    // windows, read through the span reader)
    constexpr bool PKX = SHIFTND_STEP_PK && sizeof(S) == 2 && ND == 2;
    const int phx = (-d.cx2 * 2) & 15, phg = (-d.cg2 * 2) & 15;   // uniform phases of the windows (PK)
    uint32_t xmask[5] = {0, 0, 0, 0, 0}, gmask[5] = {0, 0, 0, 0, 0};
    bool fx = false, fg = false;
    float sc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // PK: per-corner sums [row][column offset]
    if constexpr (PKX) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int hi = 2 * i + 1 <= E ? 2 * i + 1 : E;
            xmask[i] = (xm.cm[2 * i] >= 0 ? 0xffffu : 0u) | ((2 * i + 1 <= E && xm.cm[hi] >= 0) ? 0xffff0000u : 0u);
            gmask[i] = (gm.cm[2 * i] >= 0 ? 0xffffu : 0u) | ((2 * i + 1 <= E && gm.cm[hi] >= 0) ? 0xffff0000u : 0u);
        }
        // (zeros padding: a chunk whose window lies outside the row has every column masked: any dwords do)
        fx = xm.affine && (PAD == 0 || ((xm.base * 2) & 15) == phx);
        fg = gm.affine && (PAD == 0 || ((gm.base * 2) & 15) == phg);
    }
    auto window_packed = [&](const char *rowp, const ColState<E> &cst, bool fast, int ph, const uint32_t(&m)[5], bool valid, uint32_t(&t)[5]) {
        const uint32_t vm = valid ? 0xffffffffu : 0u;   // (fill rows are not staged: whatever the slot holds is masked)
        if (fast) {
            uint32_t o[6];
            lds_window6(rowp, cst.base * 2, ph, o);
            if (ph & 2) {   // uniform
#pragma unroll
                for (int i = 0; i < 5; ++i) t[i] = __builtin_amdgcn_alignbit(o[i + 1], o[i], 16) & (m[i] & vm);
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) t[i] = o[i] & (m[i] & vm);
            }
        } else {
            const uint16_t *p0 = reinterpret_cast<const uint16_t *>(rowp);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int hi = 2 * i + 1 <= E ? 2 * i + 1 : E;
                const uint32_t lo = p0[cst.cm[2 * i] > 0 ? cst.cm[2 * i] : 0];
                const uint32_t up = p0[cst.cm[hi] > 0 ? cst.cm[hi] : 0];
                t[i] = (lo | (up << 16)) & (m[i] & vm);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int vtr = tr + u * R;
    if (tr < R && vtr < Rn) {
        const int b = b0 + vtr;
        CT dw[3] = {static_cast<CT>(d.dw[0]), static_cast<CT>(d.dw[1]), static_cast<CT>(d.dw[2])};
        Chunk<S, E> res;
        // only zeros padding has rows without a source
        auto row_valid = [&](int pr, int cs) { return PAD != 0 || row_map_t<PAD>(pr, cs, S1) >= 0; };
        // corner row k of an element: bit 0 = + 1 plane (3-D), next bit = + 1 row
        auto corner_plane = [](int k) { return ND == 3 ? (k & 1) : 0; };
        auto corner_row = [](int k) { return ND == 3 ? ((k >> 1) & 1) : (k & 1); };
        // ---- grad_x ----------------------------------------------------------------------------------------------
        if constexpr (ACTIVE) {
            CT gv[NCC][E + 1];
#pragma unroll
            for (int k = 0; k < NCC; ++k) {
                const int ha = corner_plane(k), hb = corner_row(k);
                S raw[E + 1];
                if constexpr (PKX) lds_read_row_span<S, E>(tile + (NX + NG + ha * (RT + 1) + vtr + hb) * RB, pag[ha] >= 0 && row_valid(b + hb, d.cg1), gm, fg, phg, raw);
                else lds_read_row<S, E>(tile + (NX + NG + ha * (RT + 1) + vtr + hb) * RB, pag[ha] >= 0 && row_valid(b + hb, d.cg1), gm, raw);
#pragma unroll
                for (int e = 0; e <= E; ++e) gv[k][e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[1 << ND];
#pragma unroll
                for (int q = 0; q < (1 << ND); ++q) v[q] = gv[q & (NCC - 1)][e + (q >> (ND - 1))];
                res.e[e] = narrow<T>(interp_t<T, ND>(v, dw));
            }
        } else if constexpr (PK) {
            uint32_t t[5];
            window_packed(tile + (NX + vtr) * RB, gm, fg, phg, gmask, true, t);
            __builtin_memcpy(res.e, t, 16);
        } else {
            // 2-D: the staged grad_out row b, read through the column map, IS a grad_x row (which one: below);
            // 3-D: the staged row g1[b] of plane g0[a]
            S graw[E + 1];
            const bool gvalid = SCAT || (pag[0] >= 0 && row_valid(b, d.cg1));
            lds_read_row<S, E>(tile + ((SCAT ? NX : NX + NG) + vtr) * RB, gvalid, gm, graw);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = graw[e];
        }
        // ---- weight-gradient sums from the x corners and the incoming gradient --------------------------------------
        if constexpr (PKX) {
            uint32_t gq[4];
            __builtin_memcpy(gq, __builtin_assume_aligned(tile + (NX + vtr) * RB + ji * static_cast<int>(sizeof(S)), 16), 16);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                uint32_t t[5];
                window_packed(tile + (vtr + hb) * RB, xm, fx, phx, xmask, row_valid(b + hb, d.cx1), t);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[hb][0] = dot2_packed<T>(gq[i], t[i], sc[hb][0]);
                    sc[hb][1] = dot2_packed<T>(gq[i], __builtin_amdgcn_alignbit(t[i + 1], t[i], 16), sc[hb][1]);
                }
            }
        } else {
        CT xv[NCC][E + 1];
#pragma unroll
        for (int k = 0; k < NCC; ++k) {
            const int ha = corner_plane(k), hb = corner_row(k);
            S raw[E + 1];
            lds_read_row<S, E>(tile + (ha * (RT + 1) + vtr + hb) * RB, pax[ha] >= 0 && row_valid(b + hb, d.cx1), xm, raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[k][e] = widen<T>(raw[e]);
        }
        Chunk<S, E> gch;
        __builtin_memcpy(gch.e, __builtin_assume_aligned(tile + (NX + vtr) * RB + ji * static_cast<int>(sizeof(S)), 16), 16);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            CT v[1 << ND], df[NDIFF];
#pragma unroll
            for (int q = 0; q < (1 << ND); ++q) v[q] = xv[q & (NCC - 1)][e + (q >> (ND - 1))];
            corner_diffs<ND, CT>(v, df);
            const CT gval = widen<T>(gch.e[e]);
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) part[i] = fma_ct(gval, df[i], part[i]);
        }
        }
        // ---- store ---------------------------------------------------------------------------------------------------
        if constexpr (SCAT) {
            // periodic padding is a permutation (the x map is the inverse of the grad map); otherwise row b - shift when
            // that is a row, and the rows no grad_out row reaches are the tail below
            const int brow = PAD == 2 ? row_map_t<PAD>(b, d.cx1, S1) : b - d.scat;
            if (brow >= 0 && brow < S1) store_chunk<S, E>(gxp + static_cast<int64_t>(brow) * S2 + ji, res);
            if (PAD != 2 && d.scat != 0) {
                const int e0 = d.scat > 0 ? max(S1 - d.scat, 0) : 0, e1 = d.scat > 0 ? S1 : min(-d.scat, S1);
                if (b >= e0 && b < e1) {  // this row of grad_x has no source by the plain shift: fill, or a clamped / reflected row
                    const int src = row_map_t<PAD>(b, d.cg1, S1);
                    S zero;
                    __builtin_memset(&zero, 0, sizeof(S));
                    Chunk<S, E> t;
                    bool gcontig = true;
#pragma unroll
                    for (int e = 0; e < E; ++e) gcontig = gcontig && gm.cm[e] >= 0 && gm.cm[e] == gm.cm[0] + e;
                    if (src < 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) t.e[e] = zero;
                    } else if constexpr (POOL) {
                        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(src), p.d_k1));
                        const int rc = min(p.K1, S1 - pr * p.K1);
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int col = gm.cm[e] >= 0 ? gm.cm[e] : 0;
                            const int pc = static_cast<int>(fdiv(static_cast<uint32_t>(col), p.d_k2));
                            const S q = narrow<T>(div_count<CT>(widen<T>(gp[static_cast<int64_t>(pr) * p.P2 + pc]), rc * min(p.K2, S2 - pc * p.K2)));
                            t.e[e] = gm.cm[e] >= 0 ? q : zero;
                        }
                    } else if (gcontig) {
                        t = load_chunk<S, E>(gp + static_cast<int64_t>(src) * S2 + gm.cm[0]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) t.e[e] = gm.cm[e] >= 0 ? gp[static_cast<int64_t>(src) * S2 + gm.cm[e]] : zero;
                    }
                    store_chunk<S, E>(gxp + static_cast<int64_t>(b) * S2 + ji, t);
                }
            }
        } else {
            store_chunk<S, E>(gxp + static_cast<int64_t>(b) * S2 + ji, res);
        }
    }
    }
    if constexpr (PKX) {   // corner_diffs<2> of the per-corner sums (it is linear): column difference at row 0, at row 1
        part[0] = static_cast<CT>(sc[0][1] - sc[0][0]);
        part[1] = static_cast<CT>(sc[1][1] - sc[1][0]);
    }
    // ---- this step's sums: DPP tree per wave, the four waves added in fp64 by one thread ----------------------------
    double *scratch = reinterpret_cast<double *>(tile + ((npieces * 16 + 63) & ~63));
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) {
        const CT t = wave_total(part[i]);
        if ((tid & 63) == 63) scratch[NDIFF * wave + i] = static_cast<double>(t);
    }
    __syncthreads();
    if (tid < NDIFF) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) acc += scratch[NDIFF * w + tid];
        p.partials[static_cast<size_t>(bid) * NDIFF + tid] = acc;
    }
}


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
            cs0 = canon_of<PAD, double>(rint(wv[0]), p.S0, p.d_per0);
            cs1 = canon_of<PAD, double>(rint(wv[1]), p.S1, p.d_per1);
            cs2 = canon_of<PAD, double>(rint(wv[2]), p.S2, p.d_per2);
        } else {
            float wv[3];
            load_weights_nd<float>(p.w, p.wkind, c, 3, wv);
            cs0 = canon_of<PAD, float>(rintf(wv[0]), p.S0, p.d_per0);
            cs1 = canon_of<PAD, float>(rintf(wv[1]), p.S1, p.d_per1);
            cs2 = canon_of<PAD, float>(rintf(wv[2]), p.S2, p.d_per2);
        }
        cs0 = __builtin_amdgcn_readfirstlane(cs0);
        cs1 = __builtin_amdgcn_readfirstlane(cs1);
        cs2 = __builtin_amdgcn_readfirstlane(cs2);
        pa = row_map_t<PAD>(a + p.L0, cs0, p.S0);
    } else {
        plane = fdiv(bid, p.d_spp);
        step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
        const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
        channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2);
    }
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int r = step * p.R + tr;
    if (tr >= p.R || r >= p.O1) return;
    const int jo = tc * E;
    const int rb = row_map_t<PAD>(r + p.L1, cs1, p.S1);
    int mm[E];
    bool contig = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        mm[e] = row_map_t<PAD>(jo + p.L2 + e, cs2, p.S2);
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
template <typename T, int PAD>
__global__ __launch_bounds__(kThreads) void step_gather_forward_pool(const GatherParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs1, cs2;
    channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2);
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int P1 = (p.O1 + 1) >> 1, P2 = p.O2 >> 1;
    const int pr = step * p.R + tr;   // pooled row
    if (tr >= p.R || pr >= P1) return;
    const int jo = tc * E;
    int mm[E];
    bool contig = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        mm[e] = row_map_t<PAD>(jo + p.L2 + e, cs2, p.S2);
        contig = contig && (mm[e] == mm[0] + e);
    }
    contig = contig && mm[0] >= 0;
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const int n1 = min(2, p.O1 - 2 * pr);   // rows of this window row (a ragged last one: 1)
    S zero;
    __builtin_memset(&zero, 0, sizeof(S));
    Chunk<S, E> v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rb = h < n1 ? row_map_t<PAD>(2 * pr + h + p.L1, cs1, p.S1) : -1;
        if (rb < 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[h].e[e] = zero;
        } else {
            const S *row = xp + rb * p.S2;
            if (contig) {
                v[h] = load_chunk<S, E, true>(row + mm[0]);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) v[h].e[e] = mm[e] >= 0 ? __builtin_nontemporal_load(row + mm[e]) : zero;
            }
        }
    }
    Chunk<S, (E / 2 > 0 ? E / 2 : 1)> outc;
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        CT acc = (CT(0) + widen<T>(v[0].e[2 * j])) + widen<T>(v[0].e[2 * j + 1]);
        if (n1 == 2) acc = (acc + widen<T>(v[1].e[2 * j])) + widen<T>(v[1].e[2 * j + 1]);
        outc.e[j] = narrow<T>(div_count<CT>(acc, n1 * 2));
    }
    S *dst = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + static_cast<int64_t>(pr) * P2 + jo / 2;
    __builtin_memcpy(__builtin_assume_aligned(dst, 8), outc.e, 8);
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
    channel_shifts2<PAD>(p.w, p.wkind, p.wzp, c, p.S1, p.S2, p.d_per1, p.d_per2, cs1, cs2);
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int r = step * p.R + tr;
    if (tr >= p.R || r >= p.O1) return;
    const int rb = row_map_t<PAD>(r + p.L1, cs1, p.S1);
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
            const int m = row_map_t<PAD>(tc * E + p.L2 + e, cs2, p.S2);
            v.e[e] = m >= 0 ? row[m] : static_cast<R_t>(p.fill);
        }
        __builtin_memcpy(&o, v.e, 16);
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(dst));
}

// The interpolating forward of 4- / 8-byte elements in the same shape, by direct loads: per corner row one element-aligned
// 16-byte raw-buffer load for the chunk's E columns and one element load for column E.  The buffer resource is the
// (n, c) plane; a window that starts before its ROW (zeros padding: shift to the right at the row start) reads the
// neighbouring row's tail, which the column mask then discards, so with zeros padding every chunk inside the plane is one
// window; the windows that would straddle the plane's first or last byte, and the row-end chunks of the other paddings, go
// element by element.
template <typename T, int PAD>
__global__ __launch_bounds__(kThreads) void step_active_forward_direct(const GatherParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wr, wc;
    load_weights2<CT>(p.w, p.wkind, c, wr, wc);
    const CT rr = c_floor<CT>(wr), rc = c_floor<CT>(wc);  // weights_init_forward, active: floor + fraction
    const CT dw[2] = {wr - rr, wc - rc};
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr, p.S1, p.d_per1));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rc, p.S2, p.d_per2));
    const int tid = static_cast<int>(threadIdx.x);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * p.cpr;
    const int r = step * p.R + tr;
    if (tr >= p.R || r >= p.O1) return;
    const int jo = tc * E;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(static_cast<const char *>(p.x)) + static_cast<int64_t>(plane) * p.x_plane * ES, 0,
        static_cast<int>(p.x_plane * ES), 0x00020000);
    int mm[E + 1];
    bool affine = true;
    int base = 0;
    bool found = false;
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        mm[e] = row_map_t<PAD>(jo + p.L2 + e, cs2, p.S2);
        if (!found && mm[e] >= 0) {
            base = mm[e] - e;
            found = true;
        }
    }
#pragma unroll
    for (int e = 0; e <= E; ++e) affine = affine && (mm[e] < 0 || mm[e] == base + e);
    CT xv[2][E + 1];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int rb = row_map_t<PAD>(r + p.L1 + hb, cs1, p.S1);
        const int rowoff = (rb < 0 ? 0 : rb) * p.S2;
        // (the first row's windows that start before the plane, and the last row's that end behind it, take the element
        //  path: a 16-byte buffer load that straddles the resource's range is answered with zeros as a whole)
        const bool inside = rowoff + base >= 0 && rowoff + base + E + 1 <= static_cast<int>(p.x_plane);
        if ((PAD == 0 || affine) && inside) {
            const uint32_t off = static_cast<uint32_t>(rowoff + base) * ES;
            S raw[E + 1];
            if constexpr (ES == 4) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2);
                __builtin_memcpy(raw, &v, 16);
                const uint32_t t = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 16, 0, 2);
                __builtin_memcpy(&raw[E], &t, 4);
            } else {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2);
                __builtin_memcpy(raw, &v, 16);
                const u2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16, 0, 2);
                __builtin_memcpy(&raw[E], &t, 8);
            }
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[hb][e] = (rb >= 0 && mm[e] >= 0) ? widen<T>(raw[e]) : CT(0);
        } else {
            const S *row = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane + rowoff;
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[hb][e] = (rb >= 0 && mm[e] >= 0) ? widen<T>(row[mm[e]]) : CT(0);
        }
    }
    Chunk<S, E> res;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
        res.e[e] = narrow<T>(interp_t<T, 2>(v, dw));
    }
    store_chunk<S, E>(static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + r * p.O2 + jo, res);
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
    const int cs0 = ND == 3 ? __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], p.S0, p.d_per0)) : 0;
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], p.S1, p.d_per1));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], p.S2, p.d_per2));
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
    for (int h = 0; h < NPL; ++h) pa[h] = ND == 3 ? row_map_t<PAD>(a + p.L0 + h, cs0, S0) : 0;

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
                const int src = row_map_t<PAD>(b0 + p.L1 + vtr, cs1, S1);
#pragma unroll
                for (int h = 0; h < NPL; ++h)
                    if (src >= 0 && pa[h] >= 0) dma(pa[h] * S1 + src, tc, h * PR * cpr + u * R * cpr);
            }
        }
        if (ACTIVE && Rn == RT && tid < cpr) {
            const int src = row_map_t<PAD>(b0 + p.L1 + RT, cs1, S1);
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
                    int src = row_map_t<PAD>(b0 + p.L1 + slot, cs1, S1);
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
        xm = fold_colstate<E, PAD>(jo + p.L2, cs2, S2);
    }
    // window reads: two aligned 16-byte spans and the workgroup's phase (lds_window6: no bank conflicts); chunks that are not
    // affine at that phase (and cropped problems, whose staged rows start at another column) read element by element
    const int phw = ((p.L2 - cs2) * static_cast<int>(sizeof(S))) & 15;
    const bool fastw = xm.affine && ((xm.base * static_cast<int>(sizeof(S))) & 15) == phw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto row_valid = [&](int pr) { return PAD != 0 || row_map_t<PAD>(pr, cs1, S1) >= 0; };
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
    const int cs0 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], p.S0, p.d_per0));
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], p.S1, p.d_per1));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], p.S2, p.d_per2));
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
    const int src_own = own ? row_map_t<PAD>(b0 + tr, cs1, S1) : -1;
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
        xm = fold_colstate<E, PAD>(jo, cs2, S2);
    }
    // window reads: two aligned 16-byte spans and the workgroup's phase (lds_window6: no bank conflicts)
    const int phw = (-cs2 * static_cast<int>(sizeof(S))) & 15;
    const bool fastw = xm.affine && (PAD == 0 || ((xm.base * static_cast<int>(sizeof(S))) & 15) == phw);
    const bool mine = tr < R && tr < Rn;   // this thread produces a chunk
    const int b = b0 + tr;
    bool rv[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) rv[hb] = PAD != 0 || row_map_t<PAD>(b + hb, cs1, S1) >= 0;
    const int RBL = cpr * 16;  // bytes per staged row
    const char *rows = tile + tr * RBL;

    // plane map(0): the first step's "+0" rows
    CT carried[2][E + 1];
    {
        const int pa0 = row_map_t<PAD>(0, cs0, S0);
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
    u4 stA = load_plane(row_map_t<PAD>(1, cs0, S0));                            // the "+1" plane of step 0
    u4 stB = load_plane(1 < p.O0 ? row_map_t<PAD>(2, cs0, S0) : -1);            // ... of step 1
    auto walk_step = [&](int a, u4 &pend) {   // `pend`: the "+1" plane of step a; leaves with that of step a + 2 in flight
        const int pa1 = row_map_t<PAD>(a + 1, cs0, S0);
        park(pend);
        __syncthreads();
        pend = load_plane(a + 2 < p.O0 ? row_map_t<PAD>(a + 3, cs0, S0) : -1);
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
template <typename T, int PAD, bool POOL = false, bool ACTIVE = true>
__global__ __launch_bounds__(kThreads) void walk_backward(const StepParams p) {
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
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * (POOL ? p.g_plane : p.x_plane);
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane;

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_cpr)), tc = tid - tr * cpr;
    const int ji = tc * E;
    ColState<E> xm, gm;
    if constexpr (PAD == 0) {
        auto affine_state = [&](int cs) {
            ColState<E> st;
            st.base = ji - cs;
            if (st.base + E < 0 || st.base >= S2) st.base = 0;
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) st.cm[e] = (ji - cs + e >= 0 && ji - cs + e < S2) ? ji - cs + e : -1;
            return st;
        };
        xm = affine_state(d.cx2);
        gm = affine_state(d.cg2);
    } else {
        const size_t rec = (static_cast<size_t>(c) * cpr + tc) * REC;
        xm = load_colstate<E>(p.colx + rec);
        gm = load_colstate<E>(p.colg + rec);
    }
    // the pieces this thread stages, the same for every plane (see walk_forward)
    // (the host picks R with (R + 1) * cpr <= 256: thread (tr, tc), tr <= R, stages piece tc of row tr -- the "+1" corner row of
    // the step's last row included -- so a plane costs one load per tensor and thread, and two planes can be in flight)
    const bool own = tr <= R && tr <= Rn;
    const int sx_own = own ? row_map_t<PAD>(b0 + tr, d.cx1, S1) : -1;
    const int sg_own = (own && (ACTIVE || tr < R)) ? row_map_t<PAD>(b0 + tr, d.cg1, S1) : -1;
    auto piece_off = [&](int row, int piece) { return static_cast<uint32_t>(max(row, 0) * S2 + piece * E) * static_cast<uint32_t>(sizeof(S)); };
    const uint32_t ox_own = piece_off(sx_own, tc), og_own = piece_off(sg_own, tc);
    const uint32_t plane_bytes = static_cast<uint32_t>(S1) * static_cast<uint32_t>(S2) * static_cast<uint32_t>(sizeof(S));
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
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<S *>(gp), 0, POOL ? static_cast<uint32_t>(p.g_plane) * static_cast<uint32_t>(sizeof(S)) : vol_bytes, kRsrcFlags);
    // POOL: the 8 bytes of pooled row `row / K1` under piece `piece` of unpooled row `row`, bytes within a pooled plane; the
    // window rows the pooled row averages
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    auto pooled_off = [&](int row, int piece) {
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(max(row, 0)), p.d_k1));
        return static_cast<uint32_t>(pr * p.P2 + piece * (E / 2)) * static_cast<uint32_t>(sizeof(S));
    };
    auto pooled_rows = [&](int row) {
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(max(row, 0)), p.d_k1));
        return min(p.K1, S1 - pr * p.K1);
    };
    const uint32_t pooled_plane_bytes = POOL ? static_cast<uint32_t>(p.P1) * static_cast<uint32_t>(p.P2) * static_cast<uint32_t>(sizeof(S)) : 0u;
    auto pooled_plane = [&](int pa, uint32_t &soff, int &n0) {   // unpooled plane (uniform) -> byte offset of its pooled plane, window planes
        const int pp = static_cast<int>(fdiv(static_cast<uint32_t>(max(pa, 0)), p.d_k0));
        soff = static_cast<uint32_t>(pp) * pooled_plane_bytes;
        n0 = min(p.K0, S0 - pp * p.K0);
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
            v.po = __builtin_amdgcn_raw_buffer_load_b64(pag >= 0 ? gres : none, vg_own, sg, 0);
        } else {
            const uint32_t sg = pag >= 0 ? static_cast<uint32_t>(pag) * plane_bytes : 0u;
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
    const int phx = (-d.cx2 * ES) & 15, phg = (-d.cg2 * ES) & 15;
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
        const int pax0 = row_map_t<PAD>(a0, d.cx0, S0), pag0 = ACTIVE ? row_map_t<PAD>(a0, d.cg0, S0) : -1;
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
    const uint32_t myp = (POOL && mine) ? pooled_off(b, tc) : kOOR;   // POOL: its pooled bytes
    const int n1_my = POOL ? pooled_rows(b) : 1;
    auto load_own = [&](int a, bool have) {   // the incoming gradient at the thread's own chunk of plane a (raw: u4, or the pooled 8 bytes in .xy)
        u4 r;
        if constexpr (POOL) {
            uint32_t sg;
            int n0;
            pooled_plane(a, sg, n0);
            const u2 q = __builtin_amdgcn_raw_buffer_load_b64(have ? gres : none, myp, sg, 0);
            r = u4{q.x, q.y, static_cast<uint32_t>(n0), 0u};
        } else {
            r = __builtin_amdgcn_raw_buffer_load_b128(have ? gres : none, my, static_cast<uint32_t>(a) * plane_bytes, 0);
        }
        return r;
    };
    __syncthreads();   // the "+0" planes have been read
    constexpr int GA = ACTIVE ? 1 : 0;   // the gradient plane of step a: the "+1" corner plane / the plane the tap reads
    // the planes of steps a0 and a0 + 1 (steps that do not exist: empty resources; a buffer's range check does not see the
    // scalar offset)
    Staged stA, stB;
    load_planes(row_map_t<PAD>(a0 + 1, d.cx0, S0), row_map_t<PAD>(a0 + GA, d.cg0, S0), stA);
    load_planes(a0 + 1 < a1 ? row_map_t<PAD>(a0 + 2, d.cx0, S0) : -1, a0 + 1 < a1 ? row_map_t<PAD>(a0 + 1 + GA, d.cg0, S0) : -1, stB);
    u4 gcur = load_own(a0, true);
    auto walk_step = [&](int a, Staged &pend) {   // `pend` holds the planes of step a; it leaves with those of step a + 2 in flight
        park(pend);
        __syncthreads();
        const bool more = a + 2 < a1;
        load_planes(more ? row_map_t<PAD>(a + 3, d.cx0, S0) : -1, more ? row_map_t<PAD>(a + 2 + GA, d.cg0, S0) : -1, pend);
        Chunk<S, E> gch;
        if constexpr (POOL) {
            const u4 ex = expand(u2{gcur.x, gcur.y}, static_cast<int>(gcur.z) * n1_my * 2);
            __builtin_memcpy(gch.e, &ex, 16);
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
        {
            u4 bits;
            __builtin_memcpy(&bits, res.e, 16);
            __builtin_amdgcn_raw_buffer_store_b128(bits, ores, my, static_cast<uint32_t>(a) * plane_bytes, 0);
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

struct StepLayout {
    int cpr, R, U, spp, spv, rec, ndiff;
    uint64_t total_steps;
    size_t off_desc, off_colx, off_colg, bytes;
};

// row groups per thread (knob 35 bit 1 = 2: always one, bit 2 = 4: always two): two for 16-bit data, where the kernel is
// bound by instruction issue (same box, one vs two: fp16 C512 224x224 reflect 1.80 -> 1.67 ms, interpolating 1.91 -> 1.79,
// bf16 N128 C256 56x56 0.126 -> 0.109; zeros padding 1.61 vs 1.62); one for 4- / 8-byte elements (fp32 sparse 1.58 vs 1.62,
// interpolating 1.63 vs 1.67 ms: the tighter sweep front wins), 3-D and pooled calls
int step_row_groups(const Geometry &g, int es) {
    if (g.nd != 2 || g.K[0] > 0) return 1;
    if (g_step_tune[3] & 2) return 1;
    if (g_step_tune[3] & 4) return 2;
    return es == 2 ? 2 : 1;
}

// force_u: row groups per thread (0: by dtype and knob 35).  The WORKSPACE is planned with one (the most steps), so that its size
// does not depend on the thread-local knobs of whoever asks; a run lays its regions out with its own U inside that.
StepLayout step_layout(const Geometry &g, int es, int force_u = 0) {
    StepLayout L{};
    const int E = 16 / es;
    L.cpr = static_cast<int>(g.S[2] * es / 16);
    if (L.cpr < 1) L.cpr = 1;
    L.R = kThreads / L.cpr < 1 ? 1 : kThreads / L.cpr;
    if (L.R > g.S[1]) L.R = static_cast<int>(g.S[1] > 0 ? g.S[1] : 1);
    L.U = force_u > 0 ? force_u : step_row_groups(g, es);
    if (L.U * L.R > g.S[1] && L.R >= g.S[1]) L.U = 1;  // (one row group already covers the plane)
    L.spp = static_cast<int>((g.S[1] + L.U * L.R - 1) / (L.U * L.R));
    L.spv = static_cast<int>(g.S[0]) * L.spp;
    L.rec = (E + 3 <= 8) ? 8 : 16;
    L.ndiff = g.nd == 3 ? 8 : 2;
    L.total_steps = static_cast<uint64_t>(g.N) * g.C * L.spv;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    L.off_desc = up(L.total_steps * L.ndiff * sizeof(double));
    L.off_colx = L.off_desc + up(static_cast<size_t>(g.C) * sizeof(ChanDesc));
    L.off_colg = L.off_colx + up(static_cast<size_t>(g.C) * L.cpr * L.rec * sizeof(int16_t));
    L.bytes = L.off_colg + up(static_cast<size_t>(g.C) * L.cpr * L.rec * sizeof(int16_t));
    return L;
}

size_t step_lds_bytes(const StepLayout &L, int nd, bool active) {
    const int np = nd == 3 ? 2 : 1;
    const int RT = L.U * L.R;
    const int slots = np * (RT + 1) + RT + (active ? np * (RT + 1) : (nd == 3 ? RT : 0));
    return 64 + ((static_cast<size_t>(slots) * L.cpr * 16 + 63) & ~static_cast<size_t>(63)) + (kThreads / 64) * L.ndiff * sizeof(double);
}

template <typename T, int ND>
int launch_step_backward(StepParams &p, const StepLayout &L, bool active, void *gw, hipStream_t st) {
    using S = typename T::S;
    const size_t lds = step_lds_bytes(L, ND, active);
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_STEP_PAD(ACT, PADV) \
    case PADV: \
        if constexpr (ND == 2) { \
            if (p.K1 > 0) { hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV, true>), grid, block, lds, st, p); break; } \
            if (L.U == 2) { hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV, false, 2>), grid, block, lds, st, p); break; } \
        } \
        hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV>), grid, block, lds, st, p); break;
    if (active) {
        hipLaunchKernelGGL((step_prep<T, true>), dim3(p.C), block, 0, st, p);
        switch (p.pad) { SHIFTND_STEP_PAD(true, 0) SHIFTND_STEP_PAD(true, 1) SHIFTND_STEP_PAD(true, 2) SHIFTND_STEP_PAD(true, 3) default: SHIFTND_STEP_PAD(true, 4) }
    } else {
        hipLaunchKernelGGL((step_prep<T, false>), dim3(p.C), block, 0, st, p);
        switch (p.pad) { SHIFTND_STEP_PAD(false, 0) SHIFTND_STEP_PAD(false, 1) SHIFTND_STEP_PAD(false, 2) SHIFTND_STEP_PAD(false, 3) default: SHIFTND_STEP_PAD(false, 4) }
    }
#undef SHIFTND_STEP_PAD
    hipLaunchKernelGGL((step_reduce<T, ND>), dim3(p.C), block, 0, st, p, static_cast<S *>(gw));
    return SHIFTND_OK;
}

}  // namespace

void step_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 5) g_step_tune[knob] = value;
}

// contiguous 2-D / 3-D problems without crop whose rows are whole 16-byte pieces and at most one workgroup pass wide
static bool step_backward_core(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
bool walk_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);

bool step_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    return g.K[0] <= 0 && step_backward_core(g, dtype, go, x, gx);
}

// the fused shift + average-pool backward (2-D; `go` = gradient of the pooled output, contiguous)
bool step_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (!(g.K[0] > 0 && g.nd == 2)) return false;
    (void)go;
    // the interpolating shift expands three pooled pieces per thread and is faster on the band-walk kernel (N64 C256 224x224
    // fp32: 2.58 vs 2.22 ms); the sparse shift: 1.60 vs 1.65 ms, fp16 C512 1.98 vs 2.34 ms, N128 C512 56x56 0.43 vs 0.51 ms
    if (g.active && g_step_tune[0] != 2) return false;
    return step_backward_core(g, dtype, nullptr, x, gx);
}

static bool step_backward_core(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g_step_tune[0] == 1) return false;
    if (dtype > SHIFTND_BF16 || (g.nd != 2 && g.nd != 3)) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if (g.O[d] != g.S[d] || g.L[d] != 0) return false;
    if ((g.nd == 2 && g.S[0] != 1) || g.S[0] < 1 || g.S[1] < 1 || g.S[2] < 1) return false;
    if ((g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads || g.S[2] > 32000) return false;
    if (g.S[0] * g.S[1] * g.S[2] >= (1LL << 30)) return false;
    if (g.K[0] > 0) {  // pooled calls: the gradient has the pooled shape (contiguous by contract)
        if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.gs, g.N, g.C, g.S)) return false;
    } else if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O) || !dense(g.gs, g.N, g.C, g.S)) {
        return false;
    }
    if ((g.K[0] <= 0 && reinterpret_cast<uintptr_t>(go) % 16) || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(gx) % 16) return false;
    const StepLayout L = step_layout(g, es);
    if (L.total_steps + 8 >= (1ull << 31)) return false;
    if (step_lds_bytes(L, g.nd, g.active != 0) > 64 * 1024) return false;
    if (g_step_tune[0] == 2) return true;
    // 3-D: the walk through the planes where it serves (16-bit interpolating), else knob 35 bit 0
    return g.nd == 2 || (g_step_tune[3] & 1) || (g.K[0] <= 0 && walk_backward_eligible(g, dtype, go, x, gx));
}

// sparse-shift / quantized forward of 4- and 8-byte elements: dense tensors, output rows of whole 16-byte chunks and at
// most one workgroup pass wide (crops are fine: a gather)
bool step_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[1] == 1) return false;
    const int es = dtype_size(dtype);
    if (es != 1 && es != 2 && es != 4 && es != 8) return false;
    const bool interpolating = g.active && dtype <= SHIFTND_BF16;
    if (interpolating && (es < 4 || g.S[1] * g.S[2] * es >= (1LL << 31))) return false;  // (the buffer resource spans one plane)
    // 2-D; 3-D for the sparse shift of 4- / 8-byte elements (the caller checks that its weights are floats)
    if (g.nd == 3 ? (es < 4 || interpolating) : (g.nd != 2 || g.S[0] != 1 || g.O[0] != 1)) return false;
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
    // interpolating by direct loads: every corner row is loaded by two workgroups at element alignment -- measured
    // slower than the LDS-staged plane kernel (C2 tensor 1.51 vs 1.13 ms): on request only (knob 33 = 2)
    if (interpolating) return false;
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
    case 3: hipLaunchKernelGGL((step_gather_forward<ES, 3, 3>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((step_gather_forward<ES, 4, 3>), grid, block, 0, st, p); break; \
    }
        if (es == 4) { SHIFTND_STEP_FWD3(4) } else { SHIFTND_STEP_FWD3(8) }
#undef SHIFTND_STEP_FWD3
        return SHIFTND_OK;
    }
    if (g.active && dtype <= SHIFTND_BF16) {
        note_kernel("step_active_forward_direct");
#define SHIFTND_STEP_ACT(TT) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((step_active_forward_direct<TT, 0>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((step_active_forward_direct<TT, 1>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((step_active_forward_direct<TT, 2>), grid, block, 0, st, p); break; \
    case 3: hipLaunchKernelGGL((step_active_forward_direct<TT, 3>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((step_active_forward_direct<TT, 4>), grid, block, 0, st, p); break; \
    }
        if (dtype == SHIFTND_F32) { SHIFTND_STEP_ACT(f32_t) } else { SHIFTND_STEP_ACT(f64_t) }
#undef SHIFTND_STEP_ACT
        return SHIFTND_OK;
    }
    note_kernel(es < 4 ? "step_gather_forward_small" : "step_gather_forward");
#define SHIFTND_STEP_FWD(KERNEL, ES) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((KERNEL<ES, 0>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((KERNEL<ES, 1>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((KERNEL<ES, 2>), grid, block, 0, st, p); break; \
    case 3: hipLaunchKernelGGL((KERNEL<ES, 3>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((KERNEL<ES, 4>), grid, block, 0, st, p); break; \
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
    // 16-bit interpolation stays on the sliding-window kernel (0.18 vs 0.215 ms: four corner rows to unpack per output row
    // against two) unless forced (knob 34 >= 2 or knob 35 bit 3)
    if (g.nd == 3 && interpolating && es == 2 && !(g_step_tune[2] >= 2 || (g_step_tune[3] & 8))) return false;
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

template <typename T>
static void launch_step_forward_lds(const FwdParams &p, bool active, int pad, int U, size_t lds, hipStream_t st) {
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_STEP_FWD_LDS(ACT, PADV) \
    case PADV: \
        if (p.nd == 3) { \
            if (U == 2) hipLaunchKernelGGL((step_forward_lds<T, 3, ACT, PADV, 2>), grid, block, lds, st, p); \
            else hipLaunchKernelGGL((step_forward_lds<T, 3, ACT, PADV, 1>), grid, block, lds, st, p); \
        } else if (U == 2) hipLaunchKernelGGL((step_forward_lds<T, 2, ACT, PADV, 2>), grid, block, lds, st, p); \
        else hipLaunchKernelGGL((step_forward_lds<T, 2, ACT, PADV, 1>), grid, block, lds, st, p); \
        break;
    if (active) {
        switch (pad) { SHIFTND_STEP_FWD_LDS(true, 0) SHIFTND_STEP_FWD_LDS(true, 1) SHIFTND_STEP_FWD_LDS(true, 2) SHIFTND_STEP_FWD_LDS(true, 3) default: SHIFTND_STEP_FWD_LDS(true, 4) }
    } else {
        switch (pad) { SHIFTND_STEP_FWD_LDS(false, 0) SHIFTND_STEP_FWD_LDS(false, 1) SHIFTND_STEP_FWD_LDS(false, 2) SHIFTND_STEP_FWD_LDS(false, 3) default: SHIFTND_STEP_FWD_LDS(false, 4) }
    }
#undef SHIFTND_STEP_FWD_LDS
}

int step_forward_lds(const Geometry &g, int dtype, const void *x, const void *w, int wkind, uint64_t fill_bits, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    FwdParams p{};
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
    int U = g_step_tune[2] == 2 ? 1 : 2;  // (one row group: C2-tensor interpolating forward 1.15 ms, two: 1.00 ms)
    if (p.R >= p.O1) U = 1;
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
        launch_step_forward_lds<f16_t>(p, false, g.pad, U, lds, st);
        return SHIFTND_OK;
    }
    switch (dtype) {
    case SHIFTND_F32: launch_step_forward_lds<f32_t>(p, true, g.pad, U, lds, st); break;
    case SHIFTND_F64: launch_step_forward_lds<f64_t>(p, true, g.pad, U, lds, st); break;
    case SHIFTND_F16: launch_step_forward_lds<f16_t>(p, true, g.pad, U, lds, st); break;
    default: launch_step_forward_lds<bf16_t>(p, true, g.pad, U, lds, st); break;
    }
    return SHIFTND_OK;
}

// the 2-D sparse shift + 2 x 2 average pool of 4- / 8-byte float elements in one sweep of one-step workgroups
bool step_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[1] == 1) return false;
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F64) return false;
    if (g.nd != 2 || g.active || g.K[1] != 2 || g.K[2] != 2 || g.S[0] != 1 || g.O[0] != 1) return false;
    const int es = dtype_size(dtype);
    const int64_t xe = g.S[1] * g.S[2], oe = g.O[1] * g.O[2];
    if (xe < 1 || oe < 1 || xe >= (1LL << 30) || oe >= (1LL << 30)) return false;
    if ((g.O[2] * es) % 16 != 0 || g.O[2] * es / 16 > kThreads || reinterpret_cast<uintptr_t>(out) % 8 != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S)) return false;
    const int64_t cpr = g.O[2] * es / 16, R = kThreads / cpr, p1 = (g.O[1] + 1) / 2;
    return g.N * g.C * ((p1 + R - 1) / R) + 8 < (1LL << 31);
}

int step_forward_pooled(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    GatherParams p{};
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
    p.cpr = static_cast<int>(g.O[2] * es / 16);
    p.xppr = static_cast<int>(g.S[2] * es / 16);
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
#define SHIFTND_STEP_FWD_POOL(TT) \
    switch (g.pad) { \
    case 0: hipLaunchKernelGGL((step_gather_forward_pool<TT, 0>), grid, block, 0, st, p); break; \
    case 1: hipLaunchKernelGGL((step_gather_forward_pool<TT, 1>), grid, block, 0, st, p); break; \
    case 2: hipLaunchKernelGGL((step_gather_forward_pool<TT, 2>), grid, block, 0, st, p); break; \
    case 3: hipLaunchKernelGGL((step_gather_forward_pool<TT, 3>), grid, block, 0, st, p); break; \
    default: hipLaunchKernelGGL((step_gather_forward_pool<TT, 4>), grid, block, 0, st, p); break; \
    }
    if (dtype == SHIFTND_F32) { SHIFTND_STEP_FWD_POOL(f32_t) } else { SHIFTND_STEP_FWD_POOL(f64_t) }
#undef SHIFTND_STEP_FWD_POOL
    return SHIFTND_OK;
}

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
    // same box, N8 C128 16x112x112: bf16 0.178 (slide_forward) -> 0.148 ms, fp32 0.308 (step_forward_lds) -> 0.259 ms; fp64 on
    // request (knob 35 bit 5)
    return es <= 4 || (g_step_tune[3] & 32);
}

int walk_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    FwdParams p{};
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
        else hipLaunchKernelGGL((walk_forward<T, PADV, false>), grid, block, lds, st, p); \
        break;
#define SHIFTND_WALK_FWD(T) \
    switch (g.pad) { SHIFTND_WALK_FWD_PAD(T, 0) SHIFTND_WALK_FWD_PAD(T, 1) SHIFTND_WALK_FWD_PAD(T, 2) SHIFTND_WALK_FWD_PAD(T, 3) default: SHIFTND_WALK_FWD_PAD(T, 4) }
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
    if (g_step_tune[0] == 1 || (g_step_tune[3] & 16) || (g_step_tune[3] & 1)) return false;   // (bit 0: the one-step 3-D form)
    if (dtype > SHIFTND_BF16 || g.nd != 3 || g.S[0] < 2) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if (g.O[d] != g.S[d] || g.L[d] != 0) return false;
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
    // same box, N8 C128 16x112x112: bf16 0.283 (slide_backward) -> 0.263 ms, fp32 0.524 -> 0.486 ms; fp64 on request (bit 5)
    return es <= 4 || (g_step_tune[3] & 32);
}

template <typename T> static void launch_walk_backward(StepParams &p, size_t lds, bool active, void *gw, hipStream_t st) {
    using S = typename T::S;
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    if (active) hipLaunchKernelGGL((step_prep<T, true>), dim3(p.C), block, 0, st, p);
    else hipLaunchKernelGGL((step_prep<T, false>), dim3(p.C), block, 0, st, p);
#define SHIFTND_WALK_BWD(PADV) \
    case PADV: \
        if (!active && p.K0 > 0) hipLaunchKernelGGL((walk_backward<T, PADV, true, false>), grid, block, lds, st, p); \
        else if (!active) hipLaunchKernelGGL((walk_backward<T, PADV, false, false>), grid, block, lds, st, p); \
        else if (p.K0 > 0) hipLaunchKernelGGL((walk_backward<T, PADV, true>), grid, block, lds, st, p); \
        else hipLaunchKernelGGL((walk_backward<T, PADV, false>), grid, block, lds, st, p); \
        break;
    switch (p.pad) { SHIFTND_WALK_BWD(0) SHIFTND_WALK_BWD(1) SHIFTND_WALK_BWD(2) SHIFTND_WALK_BWD(3) default: SHIFTND_WALK_BWD(4) }
#undef SHIFTND_WALK_BWD
    hipLaunchKernelGGL((step_reduce<T, 3>), dim3(p.C), block, 0, st, p, static_cast<S *>(gw));
}

// the pointer-free part of step_backward_core / walk_backward_core: which geometries these kernels can serve at all
static bool step_shape_ok(const Geometry &g, int dtype) {
    if (dtype > SHIFTND_BF16 || (g.nd != 2 && g.nd != 3)) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if (g.O[d] != g.S[d] || g.L[d] != 0) return false;
    if ((g.nd == 2 && g.S[0] != 1) || g.S[0] < 1 || g.S[1] < 1 || g.S[2] < 1) return false;
    if ((g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads || g.S[2] > 32000) return false;
    return g.S[0] * g.S[1] * g.S[2] < (1LL << 30);
}

// what step_backward needs of the workspace: nothing for the geometries it never serves (crops, rows that are not whole pieces,
// rows wider than a workgroup pass: those plans used to inflate every backward workspace), and independent of the knobs
size_t step_backward_workspace(const Geometry &g, int dtype) {
    if (!step_shape_ok(g, dtype)) return 0;
    return step_layout(g, dtype_size(dtype), 1).bytes;
}

int step_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                  void *workspace, hipStream_t st) {
    const int es = dtype_size(dtype);
    const StepLayout L = step_layout(g, es);
    StepParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    char *ws = static_cast<char *>(workspace);
    p.partials = reinterpret_cast<double *>(ws);
    p.desc = reinterpret_cast<ChanDesc *>(ws + L.off_desc);
    p.colx = reinterpret_cast<int16_t *>(ws + L.off_colx);
    p.colg = reinterpret_cast<int16_t *>(ws + L.off_colg);
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.wkind = dtype;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.pad = g.pad;
    p.nd = g.nd;
    p.S0 = static_cast<int>(g.S[0]);
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.cpr = L.cpr;
    p.R = L.R;
    p.spp = L.spp;
    p.spv = L.spv;
    p.total_steps = static_cast<uint32_t>(L.total_steps);
    p.steps_per_xcd = static_cast<uint32_t>((L.total_steps + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(L.spp));
    p.d_spv = make_fastdiv(static_cast<uint32_t>(L.spv));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(L.cpr));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    if (g.K[0] > 0) {  // fused average-pool tail
        p.K1 = static_cast<int>(g.K[1]);
        p.K2 = static_cast<int>(g.K[2]);
        p.P1 = static_cast<int>(g.P[1]);
        p.P2 = static_cast<int>(g.P[2]);
        p.g_plane = g.P[1] * g.P[2];
        p.d_k1 = make_fastdiv(static_cast<uint32_t>(p.K1));
        p.d_k2 = make_fastdiv(static_cast<uint32_t>(p.K2));
    }
    if (walk_backward_eligible(g, dtype, go, x, gx) || walk_backward_pooled_eligible(g, dtype, go, x, gx)) {
        // the walk through the planes: balanced row steps, one record of sums per workgroup
        if (g.K[0] > 0) {
            p.K0 = static_cast<int>(g.K[0]);
            p.P0 = static_cast<int>(g.P[0]);
            p.d_k0 = make_fastdiv(static_cast<uint32_t>(p.K0));
            p.g_plane = g.P[0] * g.P[1] * g.P[2];
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
        note_kernel(g.K[0] > 0 ? "walk_backward_pool" : (g.active ? "walk_backward" : "walk_backward_sparse"));
        switch (dtype) {
        case SHIFTND_F32: launch_walk_backward<f32_t>(p, lds, g.active != 0, gw, st); break;
        case SHIFTND_F64: launch_walk_backward<f64_t>(p, lds, g.active != 0, gw, st); break;
        case SHIFTND_F16: launch_walk_backward<f16_t>(p, lds, g.active != 0, gw, st); break;
        default: launch_walk_backward<bf16_t>(p, lds, g.active != 0, gw, st); break;
        }
        return SHIFTND_OK;
    }
    note_kernel(g.K[0] > 0 ? "step_backward_pool" : "step_backward");
    const bool active = g.active != 0;
#define SHIFTND_STEP_T(TT) (g.nd == 3 ? launch_step_backward<TT, 3>(p, L, active, gw, st) : launch_step_backward<TT, 2>(p, L, active, gw, st))
    switch (dtype) {
    case SHIFTND_F32: return SHIFTND_STEP_T(f32_t);
    case SHIFTND_F64: return SHIFTND_STEP_T(f64_t);
    case SHIFTND_F16: return SHIFTND_STEP_T(f16_t);
    default: return SHIFTND_STEP_T(bf16_t);
    }
#undef SHIFTND_STEP_T
}

}  // namespace shiftnd
