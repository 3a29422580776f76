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
        pax[h] = ND == 3 ? row_map_t<PAD>(a + h, d.cx0, S0, p.pad) : 0;
        pag[h] = ND == 3 ? row_map_t<PAD>(a + h, d.cg0, S0, p.pad) : 0;
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
        const int sx = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cx1, S1, p.pad) : -1;  // corner rows of x: m1[b0 + tr]
#pragma unroll
        for (int h = 0; h < NP; ++h)
            if (sx >= 0 && pax[h] >= 0) dma(xp, pax[h] * S1 + sx, tc, h * (RT + 1) * cpr + u * R * cpr);
        if (vtr < Rn) {  // the incoming gradient at the rows themselves
            if constexpr (POOL) pqA = pooled_load(b0 + vtr, tc, NX * cpr + vtid);
            else dma(gp, a * S1 + b0 + vtr, tc, NX * cpr + u * R * cpr);
        }
        if constexpr (ACTIVE) {
            const int sg = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cg1, S1, p.pad) : -1;  // the rows grad_x blends: g1[b0 + tr]
            if constexpr (POOL) {
                if (sg >= 0) pqB = pooled_load(sg, tc, (NX + NG) * cpr + vtid);
            } else {
#pragma unroll
                for (int h = 0; h < NP; ++h)
                    if (sg >= 0 && pag[h] >= 0) dma(gp, pag[h] * S1 + sg, tc, (NX + NG + h * (RT + 1)) * cpr + u * R * cpr);
            }
        } else if constexpr (!SCAT) {
            const int sg = vtr < Rn ? row_map_t<PAD>(b0 + vtr, d.cg1, S1, p.pad) : -1;  // 3-D sparse shift: the one row grad_x copies
            if (sg >= 0 && pag[0] >= 0) dma(gp, pag[0] * S1 + sg, tc, (NX + NG) * cpr + u * R * cpr);
        }
    }
    }
    if (Rn == RT && tid < cpr) {  // the + 1 corner row of a full step (a ragged last step has it among its first R rows)
        const int sx = row_map_t<PAD>(b0 + RT, d.cx1, S1, p.pad);
#pragma unroll
        for (int h = 0; h < NP; ++h)
            if (sx >= 0 && pax[h] >= 0) dma(xp, pax[h] * S1 + sx, tid, (h * (RT + 1) + RT) * cpr);
        if constexpr (ACTIVE) {
            const int sg = row_map_t<PAD>(b0 + RT, d.cg1, S1, p.pad);
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
        auto row_valid = [&](int pr, int cs) { return PAD != 0 || row_map_t<PAD>(pr, cs, S1, p.pad) >= 0; };
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
            const int brow = PAD == 2 ? row_map_t<PAD>(b, d.cx1, S1, p.pad) : b - d.scat;
            if (brow >= 0 && brow < S1) store_chunk<S, E>(gxp + static_cast<int64_t>(brow) * S2 + ji, res);
            if (PAD != 2 && d.scat != 0) {
                const int e0 = d.scat > 0 ? max(S1 - d.scat, 0) : 0, e1 = d.scat > 0 ? S1 : min(-d.scat, S1);
                if (b >= e0 && b < e1) {  // this row of grad_x has no source by the plain shift: fill, or a clamped / reflected row
                    const int src = row_map_t<PAD>(b, d.cg1, S1, p.pad);
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


template <typename T, int ND>
int launch_step_backward(StepParams &p, const StepLayout &L, bool active, void *gw, hipStream_t st) {
    using S = typename T::S;
    const size_t lds = step_lds_bytes(L, ND, active);
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
    // (only the forms a call reaches are built: the fused pool for the sparse shift, two row groups per thread for 16-bit data)
#define SHIFTND_STEP_PAD(ACT, PADV) \
    case PADV: \
        if constexpr (ND == 2) { \
            if constexpr (!ACT) \
                if (p.K1 > 0) { hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV, true>), grid, block, lds, st, p); break; } \
            if constexpr (sizeof(S) == 2) \
                if (L.U == 2) { hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV, false, 2>), grid, block, lds, st, p); break; } \
        } \
        hipLaunchKernelGGL((step_backward<T, ND, ACT, PADV>), grid, block, lds, st, p); break;
    launch_step_prep(T::kDtype, active, p, st);
    if (active) {
        switch (p.pad) { SHIFTND_STEP_PAD(true, 0) SHIFTND_STEP_PAD(true, 1) SHIFTND_STEP_PAD(true, 2) default: SHIFTND_STEP_PAD(true, 3) }   // (3 = reflect and symmetric: shiftnd_step.hpp kPadMirror)
    } else {
        switch (p.pad) { SHIFTND_STEP_PAD(false, 0) SHIFTND_STEP_PAD(false, 1) SHIFTND_STEP_PAD(false, 2) default: SHIFTND_STEP_PAD(false, 3) }
    }
#undef SHIFTND_STEP_PAD
    launch_step_reduce(T::kDtype, ND, p, gw, st);
    return SHIFTND_OK;
}

}  // namespace

void launch_step_prep(int dtype, bool active, const StepParams &p, hipStream_t st) {
    const dim3 grid(p.C), block(kThreads);
#define SHIFTND_PREP(TT) hipLaunchKernelGGL((step_prep<TT>), grid, block, 0, st, p, active);
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_PREP(f32_t) break;
    case SHIFTND_F64: SHIFTND_PREP(f64_t) break;
    case SHIFTND_F16: SHIFTND_PREP(f16_t) break;
    default: SHIFTND_PREP(bf16_t) break;
    }
#undef SHIFTND_PREP
}

void launch_step_reduce(int dtype, int nd, const StepParams &p, void *grad_w, hipStream_t st) {
    const dim3 grid(p.C), block(kThreads);
#define SHIFTND_REDUCE(TT) hipLaunchKernelGGL((step_reduce<TT>), grid, block, 0, st, p, static_cast<typename TT::S *>(grad_w), nd);
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_REDUCE(f32_t) break;
    case SHIFTND_F64: SHIFTND_REDUCE(f64_t) break;
    case SHIFTND_F16: SHIFTND_REDUCE(f16_t) break;
    default: SHIFTND_REDUCE(bf16_t) break;
    }
#undef SHIFTND_REDUCE
}

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
    // fp32: 2.58 vs 2.22 ms: that form is not built); the sparse shift: 1.60 vs 1.65 ms, fp16 C512 1.98 vs 2.34 ms, N128 C512
    // 56x56 0.43 vs 0.51 ms
    if (g.active) return false;
    return step_backward_core(g, dtype, nullptr, x, gx);
}

static bool step_backward_core(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g_step_tune[0] == 1) return false;
    if (dtype > SHIFTND_BF16 || (g.nd != 2 && g.nd != 3)) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if ((g.O[d] != g.S[d] || g.L[d] != 0) && !((es == 4 || g.K[0] > 0) && walk_crop_window_ok(g, g.K[0] > 0))) return false;   // (a window: walk_backward<.., CROP>, 3-D)
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
    // 3-D: only what round 3's walk through the planes serves (shiftnd_walk3.hip; step_backward() hands those calls over).  The
    // one-step 3-D form lost to the walk and to the sliding window wherever it was measured (round 3: C3 0.43 vs 0.283 ms) and is
    // no longer instantiated.
    if (g.nd == 3) return g.K[0] <= 0 && walk_backward_eligible(g, dtype, go, x, gx);
    return true;
}

// the pointer-free part of step_backward_core / walk_backward_core: which geometries these kernels can serve at all
static bool step_shape_ok(const Geometry &g, int dtype) {
    if (dtype > SHIFTND_BF16 || (g.nd != 2 && g.nd != 3)) return false;
    const int es = dtype_size(dtype);
    // (a window: the cropped walk of 3-D fp32 volumes -- geometry only: the knob's state must not shrink a workspace)
    bool window_ok = g.nd == 3 && (es == 4 || g.K[0] > 0) && g.pad == 0 && g.L[2] <= 2;
    for (int d = 0; d < 3; ++d) window_ok = window_ok && g.O[d] >= 2 && g.L[d] >= 0 && g.L[d] + g.O[d] <= g.S[d];
    for (int d = 0; d < 3; ++d)
        if ((g.O[d] != g.S[d] || g.L[d] != 0) && !window_ok) return false;
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
    // 3-D: round 3's walk through the planes (shiftnd_walk3.hip) finishes the launch parameters and runs
    if (walk_backward_eligible(g, dtype, go, x, gx) || walk_backward_pooled_eligible(g, dtype, go, x, gx))
        return walk3_backward_launch(p, g, dtype, L.cpr, gw, st);
    note_kernel(g.K[0] > 0 ? "step_backward_pool" : "step_backward");
    const bool active = g.active != 0;
#define SHIFTND_STEP_T(TT) launch_step_backward<TT, 2>(p, L, active, gw, st)
    switch (dtype) {
    case SHIFTND_F32: return SHIFTND_STEP_T(f32_t);
    case SHIFTND_F64: return SHIFTND_STEP_T(f64_t);
    case SHIFTND_F16: return SHIFTND_STEP_T(f16_t);
    default: return SHIFTND_STEP_T(bf16_t);
    }
#undef SHIFTND_STEP_T
}

}  // namespace shiftnd
