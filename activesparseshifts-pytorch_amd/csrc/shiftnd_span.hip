// shiftnd_span.hip -- the one-step sweep for the tensors the reference's users actually have (round 4): CROPPED
// 2-D windows (every depthwise-conv emulation with padding < kernel / 2: modules/shifts.py:41-46, ops/shifts.cpp:93-135 -- the
// reference's own test script is N512 C16 64x64 cut to 62x62) and Shift1d (functional.py:7-36), whose rows can be longer than
// one workgroup pass.  The one-step kernels of shiftnd_step*.hip need rows that are whole 16-byte pieces at matching positions
// in every tensor; a cropped grad_out / output has neither (222 fp32 = 888 bytes per row, shifted by the window's corner).
//
// Same shape as step_backward -- one step per workgroup, workgroups in address order, LDS-DMA staging, one barrier, a DPP wave
// tree and one partial record per step -- with the ragged tensor addressed as the TENSOR's stream of 16-byte pieces: a staged
// row is the pieces that cover it, and remembers the byte phase of its first column.
//
//   span_prep       per channel: weight preparation (cpu/shifts_cpu.cpp:242-244), canonical shifts of the x maps over the input
//                   sizes and of the gradient maps over the WINDOW sizes (the reference pads grad_out with the cropped sizes,
//                   kernels/shifts_kernels.h:295-297, :319-324), column tables for the paddings other than zeros
//   crop_backward   2-D: grad_x (zero outside the window: shifts_kernels.h:271, :314) and the step's sums of g * corner
//                   difference; R rows of the x plane per step
//   crop_forward    2-D: a step is 256 consecutive chunks of the OUTPUT plane's byte stream (ragged output rows), the source
//                   rows they read staged whole; the chunks that straddle output rows are done once per workgroup by one wave
//   row_backward /  1-D: a step is a segment of 256 chunks of a row and stages only the columns its windows reach
//   row_forward
//   span_forward    the general form of crop_forward (source rows with ragged pieces too, slots decoded per piece, three read
//                   paths): what crop_forward does not take; 1100 instructions against 400, 4.2 against 6.1 TB/s -- kept for the
//                   inputs whose own rows are ragged (62 x 62, cut)
//   step_reduce     (shiftnd_step.hpp) the channel sums and the blends, as for step_backward
//
// Lesson of the first version (span_backward, removed): a kernel this short lives or dies by its control flow -- 2500
// instructions, 113 exec-mask regions and ~340 scalar instructions per wave kept the CU's one scalar unit busy for 0.57 us per
// workgroup of 11 KB (4.9 TB/s); the lean kernels issue ~200 vector + ~120 scalar instructions per wave (6.0 - 6.3 TB/s).
//
// Reference behaviour restated: kernels/shifts_kernels.h:156-220, :222-327, :132-154; interpolation.h:3-31.  Roofline: HBM, 2 s /
// 3 s bytes per element (the window's size for the output / the incoming gradient).
#include "shiftnd_step.hpp"

namespace shiftnd {
namespace {

constexpr int kSpanRounds = 6;   // staging rounds of 256 pieces: five slots of 258 pieces (one row per step, rows of 256 chunks) fit

struct SpanParams {
    const void *x;      // saved input [N, C, S1, S2]
    const void *go;     // incoming gradient [N, C, O1, O2]
    void *out;          // grad_x, like x
    const void *w;
    double *partials;   // [total_steps][NDIFF]
    ChanDesc *desc;     // [C]
    int16_t *colx;      // [C][cpr][REC] column state of every x chunk through the x column map
    int16_t *colg;      // ... of every x chunk's window position through the gradient column map (window coordinates)
    int64_t x_plane, g_plane;   // elements per (n, c)
    int wkind, N, C, pad, nd;
    int S1, S2, O1, O2, L1, L2;
    int S0, O0, L0;      // crop_backward3: planes of the input volume / of the window, the window's first plane
    int cpr, seg, nseg;  // 16-byte chunks per x row, chunks per column segment (<= 256), segments per row
    int R, rsteps, spp;  // rows per step, row steps per plane, steps per plane = rsteps * nseg
    int P;               // pieces per slot
    int P2;              // crop_backward<.., POOL>: elements per row of the pooled gradient (2 x 2 windows)
    int P1;              // crop_backward3<.., POOL>: rows per plane of the pooled gradient
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_spp, d_C, d_seg, d_nseg, d_P, d_per1x, d_per2x, d_per1g, d_per2g;
    FastDiv d_rsteps, d_per0x, d_per0g;   // crop_backward3
};

// ---------------------------------------------------------------------------------------------------------------------
// span_prep: one workgroup per channel
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void span_prep(const SpanParams p, const bool ACTIVE) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int REC = RecSize<E>::N;
    const int c = blockIdx.x;
    const int lead = 3 - p.nd;   // real dim r -> (plane, row, column) index r + 3 - nd
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    for (int r = 0; r < p.nd; ++r) {
        const CT wv = load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd + r);
        prep_shift_backward<CT>(wv, ACTIVE, sh[r + lead], dw[r]);
    }
    const int cx1 = canon_shift(sh[1], p.S1, p.pad, p.d_per1x), cx2 = canon_shift(sh[2], p.S2, p.pad, p.d_per2x);
    const int cg1 = canon_shift(ACTIVE ? sh[1] : -sh[1], p.O1, p.pad, p.d_per1g);
    const int cg2 = canon_shift(ACTIVE ? sh[2] : -sh[2], p.O2, p.pad, p.d_per2g);
    if (threadIdx.x == 0) {
        ChanDesc d;
        d.cx0 = d.cg0 = 0;
        if (p.nd == 3) {   // the plane maps: over the input's planes, over the window's planes
            d.cx0 = canon_shift(sh[0], p.S0, p.pad, p.d_per0x);
            d.cg0 = canon_shift(ACTIVE ? sh[0] : -sh[0], p.O0, p.pad, p.d_per0g);
        }
        d.cx1 = cx1;
        d.cg1 = cg1;
        d.cx2 = cx2;
        d.cg2 = cg2;
        d.scat = 0;
        d.pad_ = 0;
        d.dw[0] = static_cast<double>(dw[0]);
        d.dw[1] = static_cast<double>(dw[1]);
        d.dw[2] = static_cast<double>(dw[2]);
        d.pad2_ = 0.0;
        p.desc[c] = d;
    }
    if (p.pad == 0) return;  // zeros padding: the column state is two compares per entry, folded in the step kernel
    for (int j = threadIdx.x; j < p.cpr; j += kThreads) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            int cm[E + 1];
            int base = 0;
            bool found = false, affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                if (which == 0) {
                    cm[e] = row_map(j * E + e, cx2, p.S2, p.pad);
                } else {  // window coordinate of x column j E + e (the map is defined on [0, O2])
                    const int q = j * E + e - p.L2;
                    cm[e] = (q >= 0 && q <= p.O2) ? row_map(q, cg2, p.O2, p.pad) : -1;
                }
                if (!found && cm[e] >= 0) {
                    base = cm[e] - e;
                    found = true;
                }
            }
#pragma unroll
            for (int e = 0; e <= E; ++e) affine = affine && (cm[e] < 0 || cm[e] == base + e);
            int16_t *rec = (which ? p.colg : p.colx) + (static_cast<size_t>(c) * p.cpr + j) * REC;
#pragma unroll
            for (int e = 0; e <= E; ++e) rec[e] = static_cast<int16_t>(cm[e]);
            rec[E + 1] = static_cast<int16_t>(base);
            rec[E + 2] = affine ? 1 : 0;
        }
    }
}

// E + 1 elements of a source row through a column state: from the staged slot, or -- a chunk whose columns are not all among the
// staged ones (only a column segment of the row is staged when rows are longer than a workgroup pass: the chunks that wrap,
// clamp or reflect at the row ends then read elsewhere) -- element by element from memory
template <typename S, int E>
__device__ __forceinline__ void span_read(const char *lds_row, const S *mem_row, bool staged, bool valid, const ColState<E> &c, S (&raw)[E + 1]) {
    if (valid && !staged) {
        S zero;
        __builtin_memset(&zero, 0, sizeof(S));
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = c.cm[e] >= 0 ? mem_row[c.cm[e]] : zero;
        return;
    }
    lds_read_row<S, E>(lds_row, valid, c, raw);
}

// ---------------------------------------------------------------------------------------------------------------------
// crop_backward: the 2-D cropped backward in step_backward's shape and at its instruction count.  As in step_backward:
//   * thread (tr, tc) stages piece tc of row tr of every group, nothing is decoded;
//   * everything comes by LDS-DMA.  The grad_out rows are ragged -- their 16-byte cover starts anywhere and can be two pieces
//     longer than an x row -- so those groups have a thread mapping of their own (piece t mod (cpr + 2) of row t div (cpr + 2):
//     lane-linear in LDS, which is what the DMA needs);
//   * one read path (ColState), the window mask applied to the result.
// 2-D, rows of at most 254 chunks (wider cropped rows: the per-channel kernels); 1-D: row_backward.
// ---------------------------------------------------------------------------------------------------------------------
// XRAG (round 5): x rows that are not whole 16-byte pieces (62 x 62, 222 x 222 fp32: the output of a cropped shift as the next
// layer's input).  4- / 8-byte elements: the x rows are staged as covers with a phase, exactly like the grad_out rows; a thread's
// chunk is ROW-RELATIVE -- elements 4 k .. 4 k + 3 of its row, cpr = ceil(row bytes / 16) of them, the last one partial -- and
// grad_x leaves through element-aligned 16-byte stores (gfx950 global stores take any alignment; a wave's chunks are still one
// contiguous run of memory).  No chunk straddles rows: the lean in-row path everywhere (the flat-stream kernels of
// shiftnd_flat.hip, which keep aligned stores, pay for that with a per-element path: N64 C256 222x222 fp32 3.0 -> 1.9 ms).
// POOL (round 6): the module's average-pool tail (modules/shifts.py:81-89: avg_pool(kernel = stride = 2, ceil_mode) behind the cropped
// shift -- every emulate_dw with padding < kernel / 2 and stride 2; other windows keep the band-walk kernels).  `go` is the gradient of the POOLED window [P1, P2]; a row of
// the unpooled gradient is its pooled row expanded, g(r, j) = grad_pooled[r / K1][j / K2] / (window size), rounded to the storage
// type like the two-step sequence (ATen's avg_pool backward).  The thread that would have moved piece `pg` of a gradient row's cover
// loads the pooled elements under columns pg E .. pg E + E - 1 instead and writes the expanded piece into the same slot at phase 0;
// everything behind the barrier is unchanged.  (The cropped pooled backward ran the band-walk kernel before: N64 C256 224x224 cut 1/1
// pool 2 fp32 3.94 ms.)
// U (round 6): row groups per thread -- a workgroup owns U R rows, every thread stages and computes U chunks R rows apart: half the
// per-workgroup scalar work, column-state prologue and partial-sum tail per byte for the form that is bound by instruction issue (the
// interpolating shift; step_backward's U).
template <typename T, bool ACTIVE, int PAD, bool XRAG = false, bool POOL = false, int U = 1>
__global__ __launch_bounds__(kThreads) void crop_backward(const SpanParams p) {
    static_assert(!(POOL && XRAG), "the pooled form takes x rows of whole pieces");
    static_assert(U == 1 || U == 2, "one or two row groups per thread");
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    constexpr int REC = RecSize<E>::N;
    constexpr int NDIFF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;  // 64-byte pads in front and behind: see lds_read_row

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);  // XCD-contiguous step ids
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c)
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int R0 = p.R, S1 = p.S1, S2 = p.S2, O1 = p.O1, O2 = p.O2, L1 = p.L1, L2 = p.L2, cpr = p.cpr;
    const int R = U * R0;   // rows of this step (R0 per row group)
    const int b0 = step * R;
    const int Rn = min(R, S1 - b0);
    // tile: x corner rows [R + 1][cpr pieces] | grad_out at the step's own rows [R][cpr + 2] | the rows grad_x reads [R (+ 1)][cpr + 2]
    // (XRAG: the x rows are covers of cpr + 2 pieces too)
    const int RBX = (XRAG ? cpr + 2 : cpr) * 16, PG = cpr + 2, RBG = PG * 16;
    const int goff = (R + 1) * RBX, gsoff = goff + R * RBG;
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * p.g_plane;
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane;
    // grad_out as the TENSOR's stream of 16-byte pieces: this plane starts gph bytes into its first piece
    const int gph = static_cast<int>((static_cast<uint64_t>(plane) * static_cast<uint64_t>(p.g_plane) * ES) & 15u);
    const char *gp16 = reinterpret_cast<const char *>(gp) - gph;

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_seg)), tc = tid - tr * cpr;
    const int ji = tc * E;
    // ---- x corner rows: LDS-DMA, thread (tr, tc) moves piece tc of row tr; the first cpr threads the + 1 row of a full step ----
    auto dma_x = [&](int src_row, int col_piece, int lds_piece0) {
        const uint32_t off = static_cast<uint32_t>(src_row * S2 + col_piece * E) * static_cast<uint32_t>(ES);
        char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
    };
    // XRAG: this plane of x starts xph bytes into its first piece of the tensor's stream
    const int xph = XRAG ? static_cast<int>((static_cast<uint64_t>(plane) * static_cast<uint64_t>(p.x_plane) * ES) & 15u) : 0;
    auto xrow_lo = [&](int row) { return xph + row * S2 * ES; };
    if constexpr (!XRAG) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (tr < R0) {
                const int vtr = tr + u * R0;
                const int sx = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cx1, S1, p.pad) : -1;
                if (sx >= 0) dma_x(sx, tc, u * R0 * cpr);
            }
        }
        if (Rn == R && tid < cpr) {
            const int sx = row_map_t<PAD>(b0 + R, d.cx1, S1, p.pad);
            if (sx >= 0) dma_x(sx, tid, R * cpr);
        }
    }
    // ---- grad_out rows: a row's cover = pieces (lo >> 4) .. of the stream, lo = gph + row O2 ES; the slot keeps them from its
    // first byte, so column j of the row sits at (lo & 15) + j ES.  A cover has up to cpr + 2 pieces, so these groups have their
    // own thread mapping -- thread t moves piece t mod (cpr + 2) of row t div (cpr + 2): lane-linear in LDS, hence LDS-DMA as well.
    // (The host picks R with R (cpr + 2) <= 256.)
    auto row_lo = [&](int row) { return gph + row * O2 * ES; };
    auto gphase = [&](int row) { return POOL ? 0 : (row_lo(row) & 15); };   // (POOL: expanded rows are written at phase 0)
    const int PGi = cpr + 2;
    const int rg = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_P)), pg = tid - rg * PGi;   // (d_P divides by cpr + 2)
    auto dma_g = [&](int row, int piece, int lds_piece0) {   // piece `piece` of the cover of grad_out row `row`
        const int lo = row_lo(row), p0 = lo >> 4, cnt = ((lo + O2 * ES + 15) >> 4) - p0;
        if (row >= 0 && piece < cnt) {
            char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp16 + static_cast<uint32_t>(p0 + piece) * 16u),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    };
    // POOL (2 x 2 windows): the expanded piece `piece` (window columns piece E ..) of window row `row`; the loads of all of a thread's
    // pieces are issued before the first conversion.  Branch-free (a kernel this short lives or dies by its control flow: the first
    // version -- any window, three load paths, IEEE divisions -- had 1009 instructions and 46 branches against 454 / 9 of the plain
    // kernel and ran 2.15 ms where the plain cropped backward takes 1.65): ONE element-aligned 8-byte load holds the E / 2 pooled
    // elements under the piece -- its start clamped into the pooled row (host: P2 >= E / 2), the elements shifted down by the
    // clamp's distance; what lies beyond the row's last window is never used (columns >= O2 are masked by the window test) -- and
    // the window counts are 1, 2 or 4: the division is a multiplication by a power of two (div_count's bits).
    struct Pooled {
        uint64_t raw;      // the E / 2 pooled elements under the piece (fp64: the one element)
        int lg;            // log2 of the rows of the pooled row's window (0 / 1)
        int dst;           // tile piece (-1: none)
        int col;
    };
    constexpr int HP = E >= 2 ? E / 2 : 1;      // pooled elements per piece
    const int opieces = (O2 * ES + 15) >> 4;   // pieces of an expanded window row
    auto pooled_load = [&](int row, int piece, int dst) {
        Pooled q;
        q.dst = (row >= 0 && piece < opieces) ? dst : -1;
        q.col = piece;
        const int pr = max(row, 0) >> 1;
        q.lg = (O1 - 2 * pr >= 2) ? 1 : 0;
        const S *prow = gp + static_cast<int64_t>(pr) * p.P2;
        const int first = piece * HP, start = min(first, p.P2 - HP);   // (P2 >= HP: host)
        if constexpr (ES == 8) {
            q.raw = *reinterpret_cast<const uint64_t *>(prow + start);
        } else {
            uint64_t v;   // element-aligned 8 bytes (load_chunk: one global_load_dwordx2 at any alignment)
            const Chunk<S, HP> h = load_chunk<S, HP>(prow + start);
            __builtin_memcpy(&v, h.e, 8);
            q.raw = v >> (static_cast<unsigned>(min(first - start, HP - 1)) * (8u * ES));
        }
        return q;
    };
    auto pooled_store = [&](const Pooled &q) {
        Chunk<S, E> out;
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            typename raw_t<ES>::type bits = static_cast<typename raw_t<ES>::type>(ES == 8 ? q.raw : (q.raw >> (h * 8 * (ES == 8 ? 0 : ES))));
            const CT val = widen<T>(__builtin_bit_cast(S, bits));
            const int k = q.lg + ((O2 - 2 * (q.col * HP + h) >= 2) ? 1 : 0);   // log2 of the window's size
            CT scale;
            if constexpr (sizeof(CT) == 4) scale = __builtin_bit_cast(float, static_cast<uint32_t>(127 - k) << 23);
            else scale = __builtin_bit_cast(double, static_cast<uint64_t>(1023 - k) << 52);
            const S v = narrow<T>(val * scale);
            if constexpr (E >= 2) {
                out.e[2 * h] = v;
                out.e[2 * h + 1] = v;
            } else {
                out.e[0] = v;
            }
        }
        if (q.dst >= 0) __builtin_memcpy(__builtin_assume_aligned(tile + q.dst * 16, 16), out.e, 16);
    };
    auto gs_row = [&](int i, bool have) {   // the row grad_x reads at step row i (window coordinates through the row map)
        const int pr = b0 + i - L1;
        const bool dom = have && pr >= 0 && (ACTIVE ? pr <= O1 : pr < O1);
        return dom ? row_map_t<PAD>(pr, d.cg1, O1, p.pad) : -1;
    };
    if constexpr (XRAG) {   // the x corner rows as covers: thread t moves piece t mod (cpr + 2) of row t div (cpr + 2); the first threads the + 1 row
        const char *xp16 = reinterpret_cast<const char *>(xp) - xph;
        auto dma_xc = [&](int row, int piece, int lds_piece0) {
            const int lo = xrow_lo(row), p0 = lo >> 4, cnt = ((lo + S2 * ES + 15) >> 4) - p0;
            if (row >= 0 && piece < cnt) {
                char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xp16 + static_cast<uint32_t>(p0 + piece) * 16u),
                                                 (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vrg = rg + u * R0;
            dma_xc((rg < R0 && vrg <= Rn) ? row_map_t<PAD>(b0 + vrg, d.cx1, S1, p.pad) : -1, pg, u * R0 * PGi);
        }
        if (Rn == R && tid < PGi) dma_xc(row_map_t<PAD>(b0 + R, d.cx1, S1, p.pad), tid, R * PGi);
    }
    if constexpr (POOL) {
        Pooled qa[U], qb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vrg = rg + u * R0;
            const int ro = (rg < R0 && vrg < Rn && b0 + vrg - L1 >= 0 && b0 + vrg - L1 < O1) ? b0 + vrg - L1 : -1;   // the step's own rows
            qa[u] = pooled_load(ro, pg, goff / 16 + u * R0 * PGi + tid);
            qb[u] = pooled_load(gs_row(vrg, rg < R0 && (ACTIVE ? vrg <= Rn : vrg < Rn)), pg, gsoff / 16 + u * R0 * PGi + tid);
        }
        Pooled qc = qa[0];
        if constexpr (ACTIVE) qc = pooled_load((Rn == R && tid < PGi) ? gs_row(R, true) : -1, tid, gsoff / 16 + R * PGi + tid);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pooled_store(qa[u]);
            pooled_store(qb[u]);
        }
        if constexpr (ACTIVE) pooled_store(qc);
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vrg = rg + u * R0;
            const int ro = (rg < R0 && vrg < Rn && b0 + vrg - L1 >= 0 && b0 + vrg - L1 < O1) ? b0 + vrg - L1 : -1;   // the step's own rows
            dma_g(ro, pg, goff / 16 + u * R0 * PGi);
            dma_g(gs_row(vrg, rg < R0 && (ACTIVE ? vrg <= Rn : vrg < Rn)), pg, gsoff / 16 + u * R0 * PGi);
        }
        if constexpr (ACTIVE) {
            if (Rn == R && tid < PGi) dma_g(gs_row(R, true), tid, gsoff / 16 + R * PGi);   // the + 1 row of a full step
        }
    }

    // ---- the thread's chunk: column state through the x map and the gradient map (window coordinates); the window mask ----
    ColState<E> xm, gm;
    {
        auto affine_state = [&](int first, int len) {
            ColState<E> st;
            st.base = first;
            if (first + E < 0 || first >= len) st.base = 0;
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e)   // (one unsigned compare: a pair of signed ones is a scalar and of two lane masks)
                st.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(len) ? first + e : -1;
            return st;
        };
        if constexpr (PAD == 0) {
            xm = affine_state(ji - d.cx2, S2);
            gm = affine_state(ji - L2 - d.cg2, O2);
            // (a window one column wide ignores the shift, shifts_kernels.h:40-48 -- both corners read column 0, not an affine
            //  state: span_backward_eligible sends O2 == 1 with zeros padding to the per-channel kernels)
        } else {
            const size_t rec = (static_cast<size_t>(c) * cpr + (tr < R0 ? tc : 0)) * REC;
            xm = load_colstate<E>(p.colx + rec);
            gm = load_colstate<E>(p.colg + rec);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    CT part[NDIFF] = {CT(0), CT(0)};
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int vtr = tr + u * R0;   // this row group's row of the tile
    if (tr < R0 && vtr < Rn) {
        const int b = b0 + vtr;
        const CT dw[3] = {static_cast<CT>(d.dw[0]), static_cast<CT>(d.dw[1]), CT(0)};
        const bool in_row = b >= L1 && b < L1 + O1;
        auto row_valid = [&](int pr, int cs, int len) { return PAD != 0 || row_map_t<PAD>(pr, cs, len, p.pad) >= 0; };
        const S zero = static_cast<S>(0.0f);
        // zeros padding: every column state is affine (the host keeps windows one column wide, whose gradient map is not, away
        // from this kernel) -- the reader without branches
        auto read_row = [&](const char *row, bool valid, const ColState<E> &cst, S (&raw)[E + 1]) {
            if constexpr (PAD == 0) lds_read_row_affine<S, E>(row, valid, cst, raw);
            else lds_read_row<S, E>(row, valid, cst, raw);
        };
        bool inside[E];   // the chunk's positions inside the window
#pragma unroll
        for (int e = 0; e < E; ++e) inside[e] = static_cast<unsigned>(ji + e - L2) < static_cast<unsigned>(in_row ? O2 : 0);
        Chunk<S, E> res;
        // ---- grad_x -------------------------------------------------------------------------------------------------------
        if constexpr (ACTIVE) {
            CT gv[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int pr = b - L1 + hb;
                const bool dom = in_row && pr <= O1;
                const int srow = dom ? row_map_t<PAD>(pr, d.cg1, O1, p.pad) : -1;
                S raw[E + 1];
                read_row(tile + gsoff + (vtr + hb) * RBG + gphase(srow), srow >= 0, gm, raw);
#pragma unroll
                for (int e = 0; e <= E; ++e) gv[hb][e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const CT v[4] = {gv[0][e], gv[1][e], gv[0][e + 1], gv[1][e + 1]};
                res.e[e] = inside[e] ? narrow<T>(interp_t<T, 2>(v, dw)) : zero;
            }
        } else {
            const int srow = in_row ? row_map_t<PAD>(b - L1, d.cg1, O1, p.pad) : -1;
            S raw[E + 1];
            read_row(tile + gsoff + vtr * RBG + gphase(srow), srow >= 0, gm, raw);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = inside[e] ? raw[e] : zero;
        }
        // ---- weight-gradient sums: corners of x against grad_out at the chunk's own position (0 outside the window) --------
        CT xv[2][E + 1];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            S raw[E + 1];
            // (XRAG: the row's first byte sits at the phase of its SOURCE row's cover)
            const int xphase = XRAG ? (xrow_lo(max(row_map_t<PAD>(b + hb, d.cx1, S1, p.pad), 0)) & 15) : 0;
            read_row(tile + (vtr + hb) * RBX + xphase, row_valid(b + hb, d.cx1, S1), xm, raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[hb][e] = widen<T>(raw[e]);
        }
        // (the own gradient chunk: E elements at column ji - L2 of the row's slot -- clamped to the slot, masked by the window)
        const S *grow = reinterpret_cast<const S *>(tile + goff + vtr * RBG + (in_row ? gphase(b - L1) : 0)) + (in_row ? ji - L2 : 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
            CT df[NDIFF];
            corner_diffs<2, CT>(v, df);
            const S graw = grow[e];
            const CT gval = widen<T>(inside[e] ? graw : zero);
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) part[i] = fma_ct(gval, df[i], part[i]);
        }
        if constexpr (XRAG) {   // element-aligned; the row's last chunk may be partial
            S *dst = gxp + static_cast<int64_t>(b) * S2 + ji;
            if (ji + E <= S2) {
                store_chunk_unaligned<S, E>(dst, res);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (ji + e < S2) dst[e] = res.e[e];
            }
        } else {
            store_chunk<S, E>(gxp + static_cast<int64_t>(b) * S2 + ji, res);
        }
    }
    }   // (row groups)
    // ---- this step's sums: DPP tree per wave, the four waves added in fp64 by one thread ----------------------------------
    double *scratch = reinterpret_cast<double *>(tile + ((gsoff + (R + 1) * RBG + 63) & ~63) + 64);
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) {
        const CT t = wave_total(part[i]);
        if ((tid & 63) == 63) scratch[NDIFF * wave + i] = static_cast<double>(t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < NDIFF) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) acc += scratch[NDIFF * w + tid];
        p.partials[static_cast<size_t>(bid) * NDIFF + tid] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// crop_backward3 (round 6): the CROPPED 3-D backward -- Shift3d behind emulate_dw with padding < kernel / 2 (modules/shifts.py:41-46:
// cut 1 / 1 per dim), which no chunk kernel took: N8 C128 16x112x112 bf16 cut 1/1/1 ran plane_backward at 1.6 ms (sparse) / 3.1 ms
// (interpolating) where the uncropped walk takes 0.24 ms.  crop_backward's one-step shape with the depth as one more step index: a
// step is R rows of ONE plane a of grad_x; it stages the R + 1 corner rows of the TWO input planes m0[a], m0[a + 1], the step's own
// gradient rows (plane a - L0 of the window) and the rows grad_x reads -- of one gradient plane g0[a - L0] (sparse shift) or two
// (interpolating) -- as covers with a phase (a row of the window starts anywhere in the tensor's stream of 16-byte pieces).  Eight sums
// of g x corner difference per step (corner_diffs<3>: bit 0 = + 1 plane, bit 1 = + 1 row, bit 2 = + 1 column, shifts_kernels.h:58-103),
// one record per step, step_reduce<.., 3> as for the walk kernels.  Reference: kernels/shifts_kernels.h:222-327 with the window of
// ops/shifts.cpp:93-135.  x rows of whole 16-byte pieces, every dim of the volume and of the window at least 2 (host).
// ---------------------------------------------------------------------------------------------------------------------
// POOL: the 2 x 2 x 2 average pool behind the cropped shift (emulate_dw with stride 2) -- `go` is the gradient of the POOLED window
// [P0, P1, P2]; a staged gradient row is its pooled row expanded at phase 0 (crop_backward<.., POOL>'s branch-free expansion with the
// plane's window count as one more power of two), which saves ATen's pool backward and its full-size gradient tensor.
// U row groups per thread: two for the interpolating shift of 4-byte elements (same box, N8 C128 16x112x112 cut 1/1/1: fp32 0.769 -> 0.725
// ms; bf16 0.53 -> 0.66 -- the second group's registers cost it its waves -- so 16-bit keeps one), one elsewhere.
template <typename T, bool ACTIVE, int PAD, bool POOL = false>
__global__ __launch_bounds__(kThreads) void crop_backward3(const SpanParams p) {
    constexpr int U = (ACTIVE && !POOL && sizeof(typename T::S) == 4) ? 2 : 1;
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    constexpr int REC = RecSize<E>::N;
    constexpr int NDIFF = WDiff<3>::N;
    constexpr int NPG = ACTIVE ? 2 : 1;   // gradient planes grad_x reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);  // XCD-contiguous step ids
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c)
    const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spp);
    const int a = static_cast<int>(fdiv(vstep, p.d_rsteps));   // the plane of grad_x
    const int step = static_cast<int>(vstep) - a * p.rsteps;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int R0 = p.R, R = U * R0, S0 = p.S0, S1 = p.S1, S2 = p.S2, O0 = p.O0, O1 = p.O1, O2 = p.O2, L0 = p.L0, L1 = p.L1, L2 = p.L2, cpr = p.cpr;
    const int b0 = step * R;
    const int Rn = min(R, S1 - b0);
    // tile: x corner rows [2 planes][R + 1][cpr pieces] | grad_out at the step's own rows [R][cpr + 2] | the rows grad_x reads [NPG][R + 1][cpr + 2]
    const int RBX = cpr * 16, PG = cpr + 2, RBG = PG * 16;
    const int goff = 2 * (R + 1) * RBX, gsoff = goff + R * RBG;
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * p.g_plane;
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane + static_cast<int64_t>(a) * S1 * S2;
    const int gph = static_cast<int>((static_cast<uint64_t>(plane) * static_cast<uint64_t>(p.g_plane) * ES) & 15u);
    const char *gp16 = reinterpret_cast<const char *>(gp) - gph;
    // the planes (uniform): input corners m0[a], m0[a + 1]; the window's plane of this step; the gradient planes grad_x reads
    int pax[2], pag[NPG];
#pragma unroll
    for (int h = 0; h < 2; ++h) pax[h] = row_map_t<PAD>(a + h, d.cx0, S0, p.pad);
    const int ao = a - L0;
    const bool in_vol = ao >= 0 && ao < O0;   // (a plane outside the window: zero gradient, nothing counted)
#pragma unroll
    for (int h = 0; h < NPG; ++h) pag[h] = (in_vol && ao + h <= O0) ? row_map_t<PAD>(ao + h, d.cg0, O0, p.pad) : -1;

    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_seg)), tc = tid - tr * cpr;
    const int ji = tc * E;
    auto dma_x = [&](int pl, int src_row, int col_piece, int lds_piece0) {
        const uint32_t off = static_cast<uint32_t>((pl * S1 + src_row) * S2 + col_piece * E) * static_cast<uint32_t>(ES);
        char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (pax[h] < 0) continue;   // (uniform)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vtr = tr + u * R0;
            if (tr < R0) {
                const int sx = vtr <= Rn ? row_map_t<PAD>(b0 + vtr, d.cx1, S1, p.pad) : -1;
                if (sx >= 0) dma_x(pax[h], sx, tc, (h * (R + 1) + u * R0) * cpr);
            }
        }
        if (Rn == R && tid < cpr) {
            const int sx = row_map_t<PAD>(b0 + R, d.cx1, S1, p.pad);
            if (sx >= 0) dma_x(pax[h], sx, tid, (h * (R + 1) + R) * cpr);
        }
    }
    // gradient rows as covers: row `vr` of the window VOLUME (plane * O1 + row) starts vrow_lo(vr) bytes into the stream of pieces
    auto vrow_lo = [&](int vr) { return gph + vr * O2 * ES; };
    auto gphase = [&](int vr) { return POOL ? 0 : (vrow_lo(vr) & 15); };
    const int PGi = cpr + 2;
    const int rg = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_P)), pg = tid - rg * PGi;
    auto dma_g = [&](int vr, int piece, int lds_piece0) {
        const int lo = vrow_lo(vr), p0 = lo >> 4, cnt = ((lo + O2 * ES + 15) >> 4) - p0;
        if (vr >= 0 && piece < cnt) {
            char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp16 + static_cast<uint32_t>(p0 + piece) * 16u),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    };
    // POOL: the expanded piece `piece` (window columns E piece ..) of row `row` of window plane `pl`
    struct Pooled {
        uint64_t raw;   // the E / 2 pooled elements under the piece
        int lg;         // log2 of (planes x rows) of the pooled element's window
        int dst;        // tile piece (-1: none)
        int col;
    };
    constexpr int HP = E >= 2 ? E / 2 : 1;
    const int opieces = (O2 * ES + 15) >> 4;
    auto pooled_load = [&](int pl, int row, int piece, int dst) {
        Pooled q;
        q.dst = (pl >= 0 && row >= 0 && piece < opieces) ? dst : -1;
        q.col = piece;
        const int pp = max(pl, 0) >> 1, pr = max(row, 0) >> 1;
        q.lg = ((O0 - 2 * pp >= 2) ? 1 : 0) + ((O1 - 2 * pr >= 2) ? 1 : 0);
        const S *prow = gp + (static_cast<int64_t>(pp) * p.P1 + pr) * p.P2;
        const int first = piece * HP, start = max(min(first, p.P2 - HP), 0);   // (P2 >= HP: host)
        if constexpr (ES == 8) {
            q.raw = *reinterpret_cast<const uint64_t *>(prow + start);
        } else {
            uint64_t v;
            const Chunk<S, HP> h = load_chunk<S, HP>(prow + start);
            __builtin_memcpy(&v, h.e, 8);
            q.raw = v >> (static_cast<unsigned>(min(first - start, HP - 1)) * (8u * ES));
        }
        return q;
    };
    auto pooled_store = [&](const Pooled &q) {
        Chunk<S, E> out;
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            typename raw_t<ES>::type bits = static_cast<typename raw_t<ES>::type>(ES == 8 ? q.raw : (q.raw >> (h * 8 * (ES == 8 ? 0 : ES))));
            const CT val = widen<T>(__builtin_bit_cast(S, bits));
            const int kk = q.lg + ((O2 - 2 * (q.col * HP + h) >= 2) ? 1 : 0);   // log2 of the window's size
            CT scale;
            if constexpr (sizeof(CT) == 4) scale = __builtin_bit_cast(float, static_cast<uint32_t>(127 - kk) << 23);
            else scale = __builtin_bit_cast(double, static_cast<uint64_t>(1023 - kk) << 52);
            const S v = narrow<T>(val * scale);
            if constexpr (E >= 2) {
                out.e[2 * h] = v;
                out.e[2 * h + 1] = v;
            } else {
                out.e[0] = v;
            }
        }
        if (q.dst >= 0) __builtin_memcpy(__builtin_assume_aligned(tile + q.dst * 16, 16), out.e, 16);
    };
    auto gs_row = [&](int i, bool have) {   // the window row grad_x reads at step row i (through the row map), or -1
        const int pr = b0 + i - L1;
        const bool dom = have && pr >= 0 && (ACTIVE ? pr <= O1 : pr < O1);
        return dom ? row_map_t<PAD>(pr, d.cg1, O1, p.pad) : -1;
    };
    if constexpr (POOL) {
        const int ro = (in_vol && rg < R && rg < Rn && b0 + rg - L1 >= 0 && b0 + rg - L1 < O1) ? b0 + rg - L1 : -1;   // the step's own rows
        const Pooled qa = pooled_load(ao, ro, pg, goff / 16 + tid);
        const int r0 = gs_row(rg, rg < R && (ACTIVE ? rg <= Rn : rg < Rn));
        const int r1 = (ACTIVE && Rn == R && tid < PGi) ? gs_row(R, true) : -1;
        Pooled qb[NPG], qc[NPG];
#pragma unroll
        for (int h = 0; h < NPG; ++h) {
            qb[h] = pooled_load(pag[h], r0, pg, gsoff / 16 + h * (R + 1) * PGi + tid);
            qc[h] = qb[h];
            if constexpr (ACTIVE) qc[h] = pooled_load(pag[h], r1, tid, gsoff / 16 + (h * (R + 1) + R) * PGi + tid);
        }
        pooled_store(qa);
#pragma unroll
        for (int h = 0; h < NPG; ++h) {
            pooled_store(qb[h]);
            if constexpr (ACTIVE) pooled_store(qc[h]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vrg = rg + u * R0;
            const int ro = (in_vol && rg < R0 && vrg < Rn && b0 + vrg - L1 >= 0 && b0 + vrg - L1 < O1) ? ao * O1 + b0 + vrg - L1 : -1;   // the step's own rows
            dma_g(ro, pg, goff / 16 + u * R0 * PGi);
        }
#pragma unroll
        for (int h = 0; h < NPG; ++h) {
            if (pag[h] < 0) continue;   // (uniform)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int vrg = rg + u * R0;
                const int r0 = gs_row(vrg, rg < R0 && (ACTIVE ? vrg <= Rn : vrg < Rn));
                dma_g(r0 >= 0 ? pag[h] * O1 + r0 : -1, pg, gsoff / 16 + (h * (R + 1) + u * R0) * PGi);
            }
            if constexpr (ACTIVE) {
                if (Rn == R && tid < PGi) {   // the + 1 row of a full step
                    const int r1 = gs_row(R, true);
                    dma_g(r1 >= 0 ? pag[h] * O1 + r1 : -1, tid, gsoff / 16 + (h * (R + 1) + R) * PGi);
                }
            }
        }
    }
    ColState<E> xm, gm;
    {
        auto affine_state = [&](int first, int len) {
            ColState<E> st;
            st.base = first;
            if (first + E < 0 || first >= len) st.base = 0;
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) st.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(len) ? first + e : -1;
            return st;
        };
        if constexpr (PAD == 0) {
            xm = affine_state(ji - d.cx2, S2);
            gm = affine_state(ji - L2 - d.cg2, O2);
        } else {
            const size_t rec = (static_cast<size_t>(c) * cpr + (tr < R0 ? tc : 0)) * REC;
            xm = load_colstate<E>(p.colx + rec);
            gm = load_colstate<E>(p.colg + rec);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    CT part[NDIFF];
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) part[i] = CT(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int vtr = tr + u * R0;   // this row group's row of the tile
    if (tr < R0 && vtr < Rn) {
        const int b = b0 + vtr;
        const CT dw[3] = {static_cast<CT>(d.dw[0]), static_cast<CT>(d.dw[1]), static_cast<CT>(d.dw[2])};
        const bool in_row = in_vol && b >= L1 && b < L1 + O1;
        auto row_valid = [&](int pr, int cs, int len) { return PAD != 0 || row_map_t<PAD>(pr, cs, len, p.pad) >= 0; };
        const S zero = static_cast<S>(0.0f);
        auto read_row = [&](const char *row, bool valid, const ColState<E> &cst, S (&raw)[E + 1]) {
            if constexpr (PAD == 0) lds_read_row_affine<S, E>(row, valid, cst, raw);
            else lds_read_row<S, E>(row, valid, cst, raw);
        };
        bool inside[E];
#pragma unroll
        for (int e = 0; e < E; ++e) inside[e] = static_cast<unsigned>(ji + e - L2) < static_cast<unsigned>(in_row ? O2 : 0);
        Chunk<S, E> res;
        // ---- grad_x: corner k of an element: bit 0 = + 1 plane, bit 1 = + 1 row (the column corners share the E + 1 columns read) ----
        if constexpr (ACTIVE) {
            CT gv[4][E + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ha = k & 1, hb = k >> 1;
                const int pr = b - L1 + hb;
                const bool dom = in_row && pr <= O1 && pag[ha] >= 0;
                const int srow = dom ? row_map_t<PAD>(pr, d.cg1, O1, p.pad) : -1;
                S raw[E + 1];
                read_row(tile + gsoff + (ha * (R + 1) + vtr + hb) * RBG + (srow >= 0 ? gphase(pag[ha] * O1 + srow) : 0), srow >= 0, gm, raw);
#pragma unroll
                for (int e = 0; e <= E; ++e) gv[k][e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                CT v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = gv[q & 3][e + (q >> 2)];
                res.e[e] = inside[e] ? narrow<T>(interp_t<T, 3>(v, dw)) : zero;
            }
        } else {
            const int srow = (in_row && pag[0] >= 0) ? row_map_t<PAD>(b - L1, d.cg1, O1, p.pad) : -1;
            S raw[E + 1];
            read_row(tile + gsoff + vtr * RBG + (srow >= 0 ? gphase(pag[0] * O1 + srow) : 0), srow >= 0, gm, raw);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = inside[e] ? raw[e] : zero;
        }
        // ---- weight-gradient sums: the eight corners of x against grad_out at the chunk's own position (0 outside the window) -------
        CT xv[4][E + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ha = k & 1, hb = k >> 1;
            S raw[E + 1];
            read_row(tile + (ha * (R + 1) + vtr + hb) * RBX, pax[ha] >= 0 && row_valid(b + hb, d.cx1, S1), xm, raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[k][e] = widen<T>(raw[e]);
        }
        const S *grow = reinterpret_cast<const S *>(tile + goff + vtr * RBG + (in_row ? gphase(ao * O1 + b - L1) : 0)) + (in_row ? ji - L2 : 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            CT v[8], df[NDIFF];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = xv[q & 3][e + (q >> 2)];
            corner_diffs<3, CT>(v, df);
            const S graw = grow[e];
            const CT gval = widen<T>(inside[e] ? graw : zero);
#pragma unroll
            for (int i = 0; i < NDIFF; ++i) part[i] = fma_ct(gval, df[i], part[i]);
        }
        store_chunk<S, E>(gxp + static_cast<int64_t>(b) * S2 + ji, res);
    }
    }   // (row groups)
    // ---- this step's sums: DPP tree per wave, the four waves added in fp64 by the first NDIFF threads -----------------------------
    double *scratch = reinterpret_cast<double *>(tile + ((gsoff + NPG * (R + 1) * RBG + 63) & ~63) + 64);
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) {
        const CT t = wave_total(part[i]);
        if ((tid & 63) == 63) scratch[NDIFF * wave + i] = static_cast<double>(t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < NDIFF) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) acc += scratch[NDIFF * w + tid];
        p.partials[static_cast<size_t>(bid) * NDIFF + tid] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The forward of the same tensors (kernels/shifts_kernels.h:156-220; weights cuda/shifts_cuda.cu:168-183): crop_forward (2-D
// windows on source rows of whole pieces) and row_forward (1-D).  The output window has ragged rows, so a step is 256 consecutive
// 16-byte chunks of the OUTPUT plane's byte stream (planes are whole pieces; a chunk may straddle two output rows).  (Round 4's
// general form for ragged SOURCE rows, span_forward -- 1100 instructions, 4.2 TB/s -- is gone: shiftnd_flat.hip serves those.)
// ---------------------------------------------------------------------------------------------------------------------
struct SpanFwdParams {
    const void *x;
    void *out;
    const void *w;
    int64_t x_plane, o_plane;   // elements per (n, c)
    int wkind, C, nd, pad;   // pad: the padding mode (the PAD = kPadMirror instantiations read it: reflect or symmetric)
    int S1, S2, O1, O2, L1, L2;
    int S0, O0, L0, rsteps;   // crop_forward3: planes of the input volume / of the window, its first plane, row steps per output plane
    int P2;                   // row_forward<.., POOL>: elements per pooled row
    int P0, P1;               // crop_forward3_pool: planes / rows of the pooled volume
    int ocp, cps, spp;   // 16-byte chunks per output plane, chunks per step (256; 254 when only a column segment is staged), steps per plane
    int P, wholeP;       // pieces per slot; the same when whole rows are staged (0: only the columns the step reaches)
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_spp, d_C, d_O2, d_P, d_per1, d_per2;
    FastDiv d_rsteps, d_per0;   // crop_forward3
};

// ---------------------------------------------------------------------------------------------------------------------
// crop_forward: the 2-D cropped forward, lean like crop_backward (round 4's general span_forward: 1100 instructions, 45 exec-mask
// regions, 4.2 / 3.1 TB/s on C2's tensor cut by one element per side).  A step is 256 consecutive 16-byte chunks of the output plane's
// byte stream (the window's rows are ragged: 222 fp32 = 888 bytes); the x rows those chunks read -- whole pieces, at most 256
// per row -- are staged by LDS-DMA, piece q of the tile = piece (q mod cpr) of source row (q div cpr).  A chunk inside one
// output row reads its window through ColState; a chunk that straddles rows goes element by element, without branches
// (clamped index + select).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool ACTIVE, int PAD, int U = 1>
__global__ __launch_bounds__(kThreads) void crop_forward(const SpanFwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wr, wc;
    load_weights2<CT>(p.w, p.wkind, c, wr, wc);
    const CT rr = ACTIVE ? c_floor<CT>(wr) : c_rint<CT>(wr), rc = ACTIVE ? c_floor<CT>(wc) : c_rint<CT>(wc);
    const CT dw[2] = {ACTIVE ? wr - rr : CT(0), ACTIVE ? wc - rc : CT(0)};
    const int S1 = p.S1, S2 = p.S2, O2 = p.O2, L1 = p.L1, L2 = p.L2, cpr = p.P;   // (P = pieces per source row)
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr, S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rc, S2, p.d_per2, p.pad));

    // U chunks per thread (the interpolating shift: a step of 512 chunks stages 11 source rows for 9.2 output rows of 222 fp32, a step
    // of 256 seven for 4.6 -- a quarter less through the LDS-DMA path, and half the per-step scalar work)
    const int q0 = step * (U * kThreads), q1 = min(p.ocp, q0 + U * kThreads);
    const int F0 = q0 * E, F1 = q1 * E;
    const int r0 = static_cast<int>(fdiv(static_cast<uint32_t>(F0), p.d_O2)), r1 = static_cast<int>(fdiv(static_cast<uint32_t>(F1 - 1), p.d_O2));
    const int nr = r1 - r0 + 1 + (ACTIVE ? 1 : 0);   // staged source rows: those of output rows r0 .. r1 (+ the corner row)
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane;
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RB = cpr * 16;
    const int npieces = nr * cpr;
    for (int k = 0; k * kThreads < npieces; ++k) {   // (uniform trip count: one to four rounds)
        const int q = k * kThreads + tid;
        const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_P));
        const int piece = q - slot * cpr;
        const int row = q < npieces ? row_map_t<PAD>(r0 + slot + L1, cs1, S1, p.pad) : -1;
        if (row >= 0) {
            char *dst_wave = tile + (k * kThreads + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + static_cast<uint32_t>(row * S2 * ES + piece * 16)),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    }
    // ---- the thread's chunks --------------------------------------------------------------------------------------------------
    auto row_ok = [&](int slot) { return PAD != 0 || row_map_t<PAD>(r0 + slot + L1, cs1, S1, p.pad) >= 0; };
    const S zero = static_cast<S>(0.0f);   // (a value, not an object the lambdas below could take the address of: that one went to scratch)
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int q = q0 + u * kThreads + tid;
    const int e0 = q * E;
    const int r = q < q1 ? static_cast<int>(fdiv(static_cast<uint32_t>(e0), p.d_O2)) : r0;
    const int j = e0 - r * O2;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int first = j + L2 - cs2;
        xm.base = (first + E < 0 || first >= S2) ? 0 : first;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(S2) ? first + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(j + L2, cs2, S2, p.pad);
    }
    if (u == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (q < q1 && j + E <= O2) {   // the chunk lies in one output row
        Chunk<S, E> res;
        const int slot = r - r0;
        if constexpr (ACTIVE) {
            CT xv[2][E + 1];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                S raw[E + 1];
                if constexpr (PAD == 0) lds_read_row_affine<S, E>(tile + (slot + hb) * RB, row_ok(slot + hb), xm, raw);
                else lds_read_row<S, E>(tile + (slot + hb) * RB, row_ok(slot + hb), xm, raw);
#pragma unroll
                for (int e = 0; e <= E; ++e) xv[hb][e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
                res.e[e] = narrow<T>(interp_t<T, 2>(v, dw));
            }
        } else {
            S raw[E + 1];
            if constexpr (PAD == 0) lds_read_row_affine<S, E>(tile + slot * RB, row_ok(slot), xm, raw);
            else lds_read_row<S, E>(tile + slot * RB, row_ok(slot), xm, raw);
#pragma unroll
            for (int e = 0; e < E; ++e) res.e[e] = raw[e];
        }
        store_chunk<S, E>(op + e0, res);
    }
    }
    // The chunks that straddle output rows -- at most one per row boundary inside the step -- are done afterwards by the first
    // wave alone, one boundary per lane, element by element (clamped reads, then one select each).  Inside the chunk loop
    // above they cost every wave the whole element-by-element path: almost every wave of 64 chunks holds one (c2acrop: 258 ->
    // ~120 vector instructions per wave).
    // (A step of 256 chunks spans more than 64 row boundaries when the window's rows are shorter than 4 chunks -- 130 x 12
    // fp32 cut to 128 x 10 has 102 -- so the lanes loop over them.)
    if (wave != 0) return;
    for (int rb = r0 + 1 + tid; rb <= r1; rb += 64) {   // the boundary between output rows rb - 1 and rb
        const uint32_t fb = static_cast<uint32_t>(rb) * static_cast<uint32_t>(O2);   // its first flat element
        const int qb = static_cast<int>(fb / E);
        if (fb % E == 0 || qb < q0 || qb >= q1) continue;
        Chunk<S, E> res;
        const int eb = qb * E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int re = static_cast<int>(fdiv(static_cast<uint32_t>(eb + e), p.d_O2));
            const int slot = re - r0, je = eb + e - re * O2;
            const int m0 = row_map_t<PAD>(je + L2, cs2, S2, p.pad);
            auto at = [&](int sl, int m) {
                const S v = reinterpret_cast<const S *>(tile + sl * RB)[m > 0 ? m : 0];
                return (m >= 0 && row_ok(sl)) ? v : zero;
            };
            if constexpr (ACTIVE) {
                const int m1 = row_map_t<PAD>(je + L2 + 1, cs2, S2, p.pad);
                const CT v[4] = {widen<T>(at(slot, m0)), widen<T>(at(slot + 1, m0)), widen<T>(at(slot, m1)), widen<T>(at(slot + 1, m1))};
                res.e[e] = narrow<T>(interp_t<T, 2>(v, dw));
            } else {
                res.e[e] = at(slot, m0);
            }
        }
        store_chunk<S, E>(op + eb, res);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ragged_forward (round 5): the 2-D forward of 4- / 8-byte elements whose SOURCE rows are not whole 16-byte pieces (62 x 62,
// 222 x 222 fp32 ...; a window included), in crop_backward<.., XRAG>'s row-relative shape: a step is R output rows of one plane,
// thread (tr, tc) owns elements 4 tc .. 4 tc + 3 of output row tr (cpr = ceil(output row bytes / 16) chunks, the last one partial);
// the R (+ 1) source rows are staged as covers with a phase (thread t moves piece t mod (xcpr + 2) of row t div (xcpr + 2));
// the output leaves through element-aligned 16-byte stores.  No chunk straddles rows: one lean path (ColState windows).
// Reference: kernels/shifts_kernels.h:156-220.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool ACTIVE, int PAD>
__global__ __launch_bounds__(kThreads) void ragged_forward(const SpanFwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    // (2-byte elements: rows of an EVEN number of them -- the host checks -- so that every row starts at a 4-byte boundary; 2-byte-aligned
    //  16-byte stores are slow, those shapes stay with shiftnd_flat.hip)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wr, wc;
    load_weights2<CT>(p.w, p.wkind, c, wr, wc);
    const CT rr = ACTIVE ? c_floor<CT>(wr) : c_rint<CT>(wr), rc = ACTIVE ? c_floor<CT>(wc) : c_rint<CT>(wc);
    const CT dw[2] = {ACTIVE ? wr - rr : CT(0), ACTIVE ? wc - rc : CT(0)};
    const int S1 = p.S1, S2 = p.S2, O1 = p.O1, O2 = p.O2, L1 = p.L1, L2 = p.L2;
    const int cpr = p.ocp, R = p.cps, PX = p.P;   // output chunks per row, rows per step, pieces per staged source row (xcpr + 2)
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr, S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rc, S2, p.d_per2, p.pad));
    const int b0 = step * R, Rn = min(R, O1 - b0);
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane;
    const int xph = static_cast<int>((static_cast<uint64_t>(plane) * static_cast<uint64_t>(p.x_plane) * ES) & 15u);
    const char *xp16 = reinterpret_cast<const char *>(xp) - xph;
    auto xrow_lo = [&](int row) { return xph + row * S2 * ES; };
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RBX = PX * 16;
    // ---- the source rows of output rows b0 .. b0 + Rn - 1 (+ the corner row): covers of at most xcpr + 1 pieces -----------------
    auto src_row = [&](int i) { return row_map_t<PAD>(b0 + i + L1, cs1, S1, p.pad); };   // (-1: padding)
    auto dma_x = [&](int row, int piece, int lds_piece0) {
        const int lo = xrow_lo(row), p0 = lo >> 4, cnt = ((lo + S2 * ES + 15) >> 4) - p0;
        if (row >= 0 && piece < cnt) {
            char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xp16 + static_cast<uint32_t>(p0 + piece) * 16u),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    };
    {
        const int rg = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_P)), pg = tid - rg * PX;
        dma_x((rg < R && (ACTIVE ? rg <= Rn : rg < Rn)) ? src_row(rg) : -1, pg, 0);
        if constexpr (ACTIVE) {
            if (Rn == R && tid < PX) dma_x(src_row(R), tid, R * PX);   // the + 1 row of a full step
        }
    }
    // ---- the thread's chunk -----------------------------------------------------------------------------------------------------
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_O2)), tc = tid - tr * cpr;   // (d_O2 divides by cpr here)
    const int ji = tc * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int first = ji + L2 - cs2;
        xm.base = (first + E < 0 || first >= S2) ? 0 : first;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(S2) ? first + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(ji + L2, cs2, S2, p.pad);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tr >= R || tr >= Rn) return;
    auto read_row = [&](int slot, S (&raw)[E + 1]) {
        const int sr = src_row(slot);
        const char *row = tile + slot * RBX + (xrow_lo(max(sr, 0)) & 15);
        if constexpr (PAD == 0) lds_read_row_affine<S, E>(row, sr >= 0, xm, raw);
        else lds_read_row<S, E>(row, sr >= 0, xm, raw);
    };
    Chunk<S, E> res;
    if constexpr (ACTIVE) {
        CT xv[2][E + 1];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            S raw[E + 1];
            read_row(tr + hb, raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[hb][e] = widen<T>(raw[e]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
            res.e[e] = narrow<T>(interp_t<T, 2>(v, dw));
        }
    } else {
        S raw[E + 1];
        read_row(tr, raw);
#pragma unroll
        for (int e = 0; e < E; ++e) res.e[e] = raw[e];
    }
    S *dst = op + static_cast<int64_t>(b0 + tr) * O2 + ji;
    if (ji + E <= O2) {
        store_chunk_unaligned<S, E>(dst, res);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (ji + e < O2) dst[e] = res.e[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// crop_forward3 (round 6): the forward of CROPPED 3-D volumes (Shift3d behind emulate_dw with padding < kernel / 2), which ran the
// per-channel kernels: N8 C128 16x112x112 bf16 cut 1/1/1 interpolating 1.0 ms (0.76 TB/s), sparse 0.34 ms, against 0.15 ms of the
// uncropped walk.  ragged_forward's row-relative shape with the depth as one more step index: a step is R output rows of ONE output
// plane ao; the R (+ 1) source rows of the source plane m0[ao + L0] (interpolating: and of m0[ao + L0 + 1]) are staged whole (source
// rows are whole 16-byte pieces) by LDS-DMA; thread (tr, tc) owns elements E tc .. of output row tr and stores them element-aligned
// (the window's rows start anywhere; 16-bit elements: rows of an even number of elements, every row at a 4-byte boundary).
// Reference: kernels/shifts_kernels.h:156-220 with the window of ops/shifts.cpp:93-135; weights cuda/shifts_cuda.cu:168-183.
// ---------------------------------------------------------------------------------------------------------------------
// ND = 2: the same kernel for cropped 2-D windows whose output PLANES are not whole 16-byte pieces -- crop_forward's flat chunk stream
// needs them to be; 110 x 110 bf16 (N32 C256 112x112 cut 1/1) ran plane_gather_forward at 2.4 TB/s, the interpolating shift the
// flat-stream kernels at 3.4 -- one source plane, two weights, interp_t<T, 2>.
// U row groups per thread: two for the interpolating shift (4-byte elements; 2-D rows form: 16-bit too), one elsewhere (same box, U = 1 -> 2: N8 C128 16x112x112 cut 1/1/1
// fp32 interpolating 0.310 -> 0.266 ms, bf16 0.234 -> 0.243; 2-D rows form, bf16 N32 C256 112x112 cut 1/1 interpolating 0.082 -> 0.072;
// the sparse shift loses: bf16 3-D 0.173 -> 0.199, 2-D 0.062 -> 0.067).  The kernel issues as many scalar as vector instructions (381 /
// 380 per wave on bf16: weights, three canonical shifts, plane / row maps, lane-mask arithmetic) and every wave of every workgroup repeats
// the uniform part; two row groups halve the workgroups.
template <typename T, bool ACTIVE, int PAD, int ND = 3>
__global__ __launch_bounds__(kThreads) void crop_forward3(const SpanFwdParams p) {
    constexpr int U = (ACTIVE && (sizeof(typename T::S) == 4 || ND == 2)) ? 2 : 1;   // (bf16 3-D interpolating: 0.234 -> 0.243-0.252 with two: one)
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    constexpr int NP = (ACTIVE && ND == 3) ? 2 : 1;   // source planes of a step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c)
    const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spp);
    const int ao = static_cast<int>(fdiv(vstep, p.d_rsteps));   // the output plane
    const int step = static_cast<int>(vstep) - ao * p.rsteps;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wv[3] = {CT(0), CT(0), CT(0)};   // the weights of the plane, row and column dims
    if constexpr (ND == 3) {
        const int wcol[3] = {0, 1, 2};
        load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 3, wcol, wv);
    } else {
        load_weights2<CT>(p.w, p.wkind, c, wv[1], wv[2]);
    }
    CT rr[3], dw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rr[k] = ACTIVE ? c_floor<CT>(wv[k]) : c_rint<CT>(wv[k]);
        dw[k] = ACTIVE ? wv[k] - rr[k] : CT(0);
    }
    const int S0 = p.S0, S1 = p.S1, S2 = p.S2, O1 = p.O1, O2 = p.O2, L0 = p.L0, L1 = p.L1, L2 = p.L2;
    const int cpr = p.ocp, R0 = p.cps, PX = p.P;   // output chunks per row, rows per row group, pieces per source row
    const int R = U * R0;                           // rows per step
    const int cs0 = ND == 3 ? __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], S0, p.d_per0, p.pad)) : 0;
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], S2, p.d_per2, p.pad));
    const int b0 = step * R, Rn = min(R, O1 - b0);
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + static_cast<int64_t>(ao) * O1 * O2;
    int pl[NP];   // source planes (uniform; -1: padding)
#pragma unroll
    for (int h = 0; h < NP; ++h) pl[h] = ND == 3 ? row_map_t<PAD>(ao + L0 + h, cs0, S0, p.pad) : 0;
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RBX = PX * 16, TP = (R + 1) * PX;   // bytes per staged row, pieces per staged plane
    auto src_row = [&](int i) { return row_map_t<PAD>(b0 + i + L1, cs1, S1, p.pad); };   // (-1: padding)
    auto dma_x = [&](int plx, int row, int piece, int lds_piece0) {
        if (row >= 0) {
            char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
            const uint32_t off = static_cast<uint32_t>((plx * S1 + row) * S2 * ES + piece * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    };
    {
        const int rg = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_P)), pg = tid - rg * PX;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int vrg = rg + u * R0;
            const int sr = (rg < R0 && (ACTIVE ? vrg <= Rn : vrg < Rn)) ? src_row(vrg) : -1;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                if (pl[h] < 0) continue;   // (uniform)
                dma_x(pl[h], sr, pg, h * TP + u * R0 * PX);
            }
        }
        if constexpr (ACTIVE) {
#pragma unroll
            for (int h = 0; h < NP; ++h)
                if (pl[h] >= 0 && Rn == R && tid < PX) dma_x(pl[h], src_row(R), tid, h * TP + R * PX);   // the + 1 row of a full step
        }
    }
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_O2)), tc = tid - tr * cpr;   // (d_O2 divides by cpr here)
    const int ji = tc * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int first = ji + L2 - cs2;
        xm.base = (first + E < 0 || first >= S2) ? 0 : first;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(S2) ? first + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(ji + L2, cs2, S2, p.pad);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto read_row = [&](int h, int slot, S (&raw)[E + 1]) {
        const bool valid = pl[h] >= 0 && src_row(slot) >= 0;
        const char *row = tile + (h * TP + slot * PX) * 16;
        if constexpr (PAD == 0) lds_read_row_affine<S, E>(row, valid, xm, raw);
        else lds_read_row<S, E>(row, valid, xm, raw);
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int vtr = tr + u * R0;
    if (tr >= R0 || vtr >= Rn) continue;
    Chunk<S, E> res;
    if constexpr (ACTIVE && ND == 3) {
        CT xv[4][E + 1];   // corner k: bit 0 = + 1 plane, bit 1 = + 1 row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            S raw[E + 1];
            read_row(k & 1, vtr + (k >> 1), raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[k][e] = widen<T>(raw[e]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            CT v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = xv[q & 3][e + (q >> 2)];
            res.e[e] = narrow<T>(interp_t<T, 3>(v, dw));
        }
    } else if constexpr (ACTIVE) {
        CT xv[2][E + 1];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            S raw[E + 1];
            read_row(0, vtr + hb, raw);
#pragma unroll
            for (int e = 0; e <= E; ++e) xv[hb][e] = widen<T>(raw[e]);
        }
        const CT dw2[2] = {dw[1], dw[2]};
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const CT v[4] = {xv[0][e], xv[1][e], xv[0][e + 1], xv[1][e + 1]};
            res.e[e] = narrow<T>(interp_t<T, 2>(v, dw2));
        }
    } else {
        S raw[E + 1];
        read_row(0, vtr, raw);
#pragma unroll
        for (int e = 0; e < E; ++e) res.e[e] = raw[e];
    }
    S *dst = op + static_cast<int64_t>(b0 + vtr) * O2 + ji;
    if (ji + E <= O2) {
        store_chunk_unaligned<S, E>(dst, res);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (ji + e < O2) dst[e] = res.e[e];
    }
    }
    (void)RBX;
}

// ---------------------------------------------------------------------------------------------------------------------
// crop_forward3_pool (round 6): Shift3d + avg_pool3d(kernel = stride = 2, ceil_mode) in one pass, cropped or not -- crop_forward3's
// shape with POOLED rows: a step is R pooled rows of one pooled plane pa; it stages the 2 R (+ 1) source rows of the two (interpolating:
// three) source planes under the window planes 2 pa, 2 pa + 1; thread (tr, tc) produces the up to four shift outputs (2 planes x 2 rows)
// of E columns -- rounded to the storage type like the unfused sequence's shift output -- sums their windows in ATen's order (plane, row,
// column) and stores E / 2 pooled elements.  p.O0 / O1 / O2: the (virtual) shift output; `out`: the pooled volume [P0, P1, P2].
// The band-walk kernel ran N8 C128 16x112x112 bf16 cut 1/1/1 pool 2 at 0.34 ms (1.3 TB/s).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool ACTIVE, int PAD>
__global__ __launch_bounds__(kThreads) void crop_forward3_pool(const SpanFwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    static_assert(E >= 2, "2- / 4-byte elements");
    constexpr int NPS = ACTIVE ? 3 : 2;   // source planes of a step
    constexpr int XR = ACTIVE ? 1 : 0;    // the + 1 corner row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c)
    const uint32_t vstep = bid - plane * static_cast<uint32_t>(p.spp);
    const int pa = static_cast<int>(fdiv(vstep, p.d_rsteps));   // the pooled plane
    const int step = static_cast<int>(vstep) - pa * p.rsteps;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wv[3];
    {
        const int wcol[3] = {0, 1, 2};
        load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 3, wcol, wv);
    }
    CT rr[3], dw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rr[k] = ACTIVE ? c_floor<CT>(wv[k]) : c_rint<CT>(wv[k]);
        dw[k] = ACTIVE ? wv[k] - rr[k] : CT(0);
    }
    const int S0 = p.S0, S1 = p.S1, S2 = p.S2, O0 = p.O0, O1 = p.O1, O2 = p.O2, L0 = p.L0, L1 = p.L1, L2 = p.L2;
    const int cpr = p.ocp, R = p.cps, PX = p.P;   // chunks per virtual output row, POOLED rows per step, pieces per source row
    const int cs0 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[0], S0, p.d_per0, p.pad));
    const int cs1 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[1], S1, p.d_per1, p.pad));
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rr[2], S2, p.d_per2, p.pad));
    const int pr0 = step * R, Rn = min(R, p.P1 - pr0);
    const int NR = 2 * R + XR;                                   // staged rows per plane (slots)
    const int nrows = min(2 * Rn, O1 - 2 * pr0) + XR;            // ... of which this step needs the first `nrows`
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    int pl[NPS];   // source planes under the window planes 2 pa (+ 1 (+ 2)); -1: padding, or beyond what the window's planes need
    const int n0 = min(2, O0 - 2 * pa);   // planes of this pooled plane's windows
#pragma unroll
    for (int k = 0; k < NPS; ++k) pl[k] = k < n0 + XR ? row_map_t<PAD>(2 * pa + k + L0, cs0, S0, p.pad) : -1;
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TP = NR * PX;   // pieces per staged plane
    auto src_row = [&](int j) { return row_map_t<PAD>(2 * pr0 + j + L1, cs1, S1, p.pad); };   // (-1: padding)
    {
        const int rg = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_P)), pg = tid - rg * PX;
        const int sr = rg < nrows ? src_row(rg) : -1;
#pragma unroll
        for (int k = 0; k < NPS; ++k) {
            if (pl[k] < 0 || sr < 0) continue;
            char *dst_wave = tile + (k * TP + wave * 64) * 16;
            const uint32_t off = static_cast<uint32_t>((pl[k] * S1 + sr) * S2 * ES + pg * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + off),
                                             (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
        }
    }
    const int tr = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_O2)), tc = tid - tr * cpr;   // (d_O2 divides by cpr here)
    const int ji = tc * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int first = ji + L2 - cs2;
        xm.base = (first + E < 0 || first >= S2) ? 0 : first;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(S2) ? first + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(ji + L2, cs2, S2, p.pad);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tr >= R || tr >= Rn) return;
    const int pr = pr0 + tr;
    const int n1 = min(2, O1 - 2 * pr);   // rows of this pooled row's windows
    auto read_row = [&](int k, int slot, S (&raw)[E + 1]) {
        const bool valid = pl[k] >= 0 && slot < nrows && src_row(slot) >= 0;
        const char *row = tile + (k * TP + slot * PX) * 16;
        if constexpr (PAD == 0) lds_read_row_affine<S, E>(row, valid, xm, raw);
        else lds_read_row<S, E>(row, valid, xm, raw);
    };
    CT acc[E / 2];
#pragma unroll
    for (int j = 0; j < E / 2; ++j) acc[j] = CT(0);
#pragma unroll
    for (int ha = 0; ha < 2; ++ha) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if (ha >= n0 || hb >= n1) continue;   // (a ragged last window: the plane / row does not exist)
            const int slot = 2 * tr + hb;
            CT y[E];   // the shift's output at window plane 2 pa + ha, row 2 pr + hb, columns ji ..
            if constexpr (ACTIVE) {
                CT xv[4][E + 1];   // corner k: bit 0 = + 1 plane, bit 1 = + 1 row
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    S raw[E + 1];
                    read_row(ha + (k & 1), slot + (k >> 1), raw);
#pragma unroll
                    for (int e = 0; e <= E; ++e) xv[k][e] = widen<T>(raw[e]);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    CT v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = xv[q & 3][e + (q >> 2)];
                    y[e] = widen<T>(narrow<T>(interp_t<T, 3>(v, dw)));
                }
            } else {
                S raw[E + 1];
                read_row(ha, slot, raw);
#pragma unroll
                for (int e = 0; e < E; ++e) y[e] = widen<T>(raw[e]);
            }
#pragma unroll
            for (int j = 0; j < E / 2; ++j) {
                acc[j] = acc[j] + y[2 * j];
                if (ji + 2 * j + 1 < O2) acc[j] = acc[j] + y[2 * j + 1];
            }
        }
    }
    S pooled[E / 2];
#pragma unroll
    for (int j = 0; j < E / 2; ++j) pooled[j] = narrow<T>(div_count<CT>(acc[j], n0 * n1 * ((ji + 2 * j + 1 < O2) ? 2 : 1)));
    S *dst = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + (static_cast<int64_t>(pa) * p.P1 + pr) * p.P2 + ji / 2;
    if (ji + E <= O2 + 1 && (ES >= 4 || (p.P2 & 1) == 0)) {   // all E / 2 pooled elements exist, the row at a 4-byte boundary
        typedef typename vec_of<8>::type v8 __attribute__((aligned(4)));
        typename vec_of<8>::type bits;
        __builtin_memcpy(&bits, pooled, 8);
        *reinterpret_cast<v8 *>(dst) = bits;
    } else {
#pragma unroll
        for (int j = 0; j < E / 2; ++j)
            if (ji + 2 * j < O2) dst[j] = pooled[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// row_forward / row_backward: Shift1d (functional.py:7-36) in the same lean shape.  A plane is one row; a step is a segment
// of 256 chunks of it and stages only the source columns its windows reach (at most 258 pieces per tensor: 256 by thread t,
// the rest by the first threads in a second DMA).  Chunks whose columns are not all among the staged ones -- the row ends of
// the wrapping / clamping / reflecting paddings -- read element by element from memory.
// ---------------------------------------------------------------------------------------------------------------------
// POOL (round 6): Shift1d + avg_pool1d(kernel = stride = 2, ceil_mode) in one pass -- the chunk's E shifted elements (rounded to the
// storage type like the unfused sequence's shift output) are summed pairwise in ATen's order and E / 2 pooled elements leave; `out`
// is the pooled row [P2], p.O2 the width of the (virtual) shift output.
template <typename T, bool ACTIVE, int PAD, bool POOL = false>
__global__ __launch_bounds__(kThreads) void row_forward(const SpanFwdParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);
    const int step = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    CT wv[3];
    load_weights_nd<CT>(p.w, p.wkind, c, 1, wv);
    const CT rc = ACTIVE ? c_floor<CT>(wv[2]) : c_rint<CT>(wv[2]);
    const CT dw[1] = {ACTIVE ? wv[2] - rc : CT(0)};
    const int S2 = p.S2, L2 = p.L2;
    const int cs2 = __builtin_amdgcn_readfirstlane(canon_of<PAD, CT>(rc, S2, p.d_per2, p.pad));
    const int q0 = step * kThreads, q1 = min(p.ocp, q0 + kThreads);
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int a0 = clampi(q0 * E + L2 - cs2, 0, S2), a1 = clampi(q1 * E + L2 - cs2 + (ACTIVE ? 1 : 0), 0, S2);   // staged columns [a0, a1)
    const int plo = (a0 * ES) >> 4, np = a1 > a0 ? ((a1 * ES + 15) >> 4) - plo : 0;
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    S *op = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane;
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma = [&](int piece, int lds_piece0) {
        char *dst_wave = tile + (lds_piece0 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(xp) + static_cast<uint32_t>(plo + piece) * 16u),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
    };
    if (tid < np) dma(tid, 0);
    if (wave == 0 && tid + kThreads < np) dma(tid + kThreads, kThreads);
    const int q = q0 + tid;
    const int j = q * E;
    ColState<E> xm;
    if constexpr (PAD == 0) {
        const int first = j + L2 - cs2;
        xm.base = (first + E < 0 || first >= S2) ? a0 : first;
        xm.affine = true;
#pragma unroll
        for (int e = 0; e <= E; ++e) xm.cm[e] = static_cast<unsigned>(first + e) < static_cast<unsigned>(S2) ? first + e : -1;
    } else {
        xm = fold_colstate<E, PAD>(j + L2, cs2, S2, p.pad);
    }
    bool staged = xm.affine;
    if constexpr (PAD != 0) {
#pragma unroll
        for (int e = 0; e <= E; ++e) staged = staged && (xm.cm[e] >= a0 && xm.cm[e] < a1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (q >= q1) return;
    S raw[E + 1];
    if (PAD == 0 || staged) {
        if constexpr (PAD == 0) lds_read_row_affine<S, E>(tile - plo * 16, true, xm, raw);   // (zeros padding: every state is affine)
        else lds_read_row<S, E>(tile - plo * 16, true, xm, raw);
    } else {
        const S zero = static_cast<S>(0.0f);
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = xm.cm[e] >= 0 ? xp[xm.cm[e]] : zero;
    }
    Chunk<S, E> res;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (ACTIVE) {
            const CT v[2] = {widen<T>(raw[e]), widen<T>(raw[e + 1])};
            res.e[e] = narrow<T>(interp_t<T, 1>(v, dw));
        } else {
            res.e[e] = raw[e];
        }
    }
    if constexpr (POOL) {
        constexpr int HP = E >= 2 ? E / 2 : 1;
        S pooled[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            const bool two = E >= 2 && j + 2 * k + 1 < p.O2;
            CT acc = CT(0) + widen<T>(res.e[E >= 2 ? 2 * k : 0]);
            if (two) acc = acc + widen<T>(res.e[E >= 2 ? 2 * k + 1 : 0]);
            pooled[k] = narrow<T>(div_count<CT>(acc, two ? 2 : 1));
        }
        S *dst = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + j / 2;
        static_assert(E >= 2, "the pooled row forward: 2-, 4-byte elements");
        if (j + E <= p.O2 + 1 && (ES >= 4 || (p.P2 & 1) == 0)) {   // all E / 2 pooled elements exist, the row at a 4-byte boundary
            typedef typename vec_of<8>::type v8 __attribute__((aligned(4)));
            typename vec_of<8>::type bits;
            __builtin_memcpy(&bits, pooled, 8);
            *reinterpret_cast<v8 *>(dst) = bits;
        } else {
#pragma unroll
            for (int k = 0; k < HP; ++k)
                if (j + 2 * k < p.O2) dst[k] = pooled[k];
        }
        return;
    }
    // (round 6) output rows that are not whole 16-byte pieces -- L4096 cut 1/1 fp32 ran the strided fallback, 4.1 ms against 0.7 --
    // leave through element-aligned stores (rows at 4-byte boundaries: host), the row's last chunk element by element
    if ((p.O2 * ES) % 16 == 0) {   // (uniform)
        store_chunk<S, E>(op + j, res);
    } else if (j + E <= p.O2) {
        store_chunk_unaligned<S, E>(op + j, res);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (j + e < p.O2) op[j + e] = res.e[e];
    }
}

// POOL (round 6): Shift1d behind a stride-2 depthwise emulation -- `go` is the gradient of the POOLED row [P2]; the two gradient spans
// are staged EXPANDED (g(j) = grad_pooled[j / 2] / window size, rounded to the storage type), piece k of a span = columns E k .. of
// the window at phase 0, by the thread that would have moved the piece (crop_backward<.., POOL>'s branch-free expansion).  N256 C512
// L4096 pool 2 ran the band-walk kernel at 1.9 - 2.5 ms against 1.07 ms of the unpooled row_backward.
template <typename T, bool ACTIVE, int PAD, bool POOL = false>
__global__ __launch_bounds__(kThreads) void row_backward(const SpanParams p) {
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int ES = sizeof(S);
    constexpr int E = 16 / ES;
    constexpr int REC = RecSize<E>::N;
    constexpr int SLOT = (kThreads + 3) * 16;   // bytes per staged span (258 pieces + one of slack)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem + 64;

    const uint32_t bid = (blockIdx.x & 7u) * p.steps_per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_steps) return;
    const uint32_t plane = fdiv(bid, p.d_spp);   // (n, c): one row
    const int sg = static_cast<int>(bid - plane * static_cast<uint32_t>(p.spp));
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    const ChanDesc d = p.desc[c];
    const int S2 = p.S2, O2 = p.O2, L2 = p.L2, cpr = p.cpr;
    const int J0 = sg * kThreads * E, J1 = min(S2, J0 + kThreads * E);   // the segment's columns of the x row
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const S *xp = static_cast<const S *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    const S *gp = static_cast<const S *>(p.go) + static_cast<int64_t>(plane) * p.g_plane;
    S *gxp = static_cast<S *>(p.out) + static_cast<int64_t>(plane) * p.x_plane;
    const int gph = static_cast<int>((static_cast<uint64_t>(plane) * static_cast<uint64_t>(p.g_plane) * ES) & 15u);
    const char *gp16 = reinterpret_cast<const char *>(gp) - gph;
    // staged columns of the three spans and their first pieces (x: row-relative; grad_out: relative to gp16)
    const int xa0 = clampi(J0 - d.cx2, 0, S2), xa1 = clampi(J1 - d.cx2 + 1, 0, S2);
    const int oa0 = clampi(J0 - L2, 0, O2), oa1 = clampi(J1 - L2, 0, O2);
    const int sa0 = clampi(J0 - L2 - d.cg2, 0, O2), sa1 = clampi(J1 - L2 - d.cg2 + (ACTIVE ? 1 : 0), 0, O2);
    const int xlo = (xa0 * ES) >> 4, xn = xa1 > xa0 ? ((xa1 * ES + 15) >> 4) - xlo : 0;
    // (POOL: the spans are pieces of the EXPANDED window row, which starts at phase 0)
    const int gph_s = POOL ? 0 : gph;
    const int olo = (gph_s + oa0 * ES) >> 4, on = oa1 > oa0 ? ((gph_s + oa1 * ES + 15) >> 4) - olo : 0;
    const int slo = (gph_s + sa0 * ES) >> 4, sn = sa1 > sa0 ? ((gph_s + sa1 * ES + 15) >> 4) - slo : 0;
    const int tid = static_cast<int>(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma = [&](const char *base16, int piece, int lds_byte0) {
        char *dst_wave = tile + lds_byte0 + wave * 64 * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base16 + static_cast<uint32_t>(piece) * 16u),
                                         (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
    };
    // POOL: the expanded piece `piece` (window columns E piece ..): one element-aligned 8-byte load of the pooled elements under it,
    // its start clamped into the pooled row (host: P2 >= E / 2), the window counts 1 or 2 (a multiplication by a power of two)
    constexpr int HP = E >= 2 ? E / 2 : 1;
    auto pooled_scale = [&](int pc) {   // 1 / (columns of pooled column pc's window)
        const int k = (O2 - 2 * pc >= 2) ? 1 : 0;
        CT scale;
        if constexpr (sizeof(CT) == 4) scale = __builtin_bit_cast(float, static_cast<uint32_t>(127 - k) << 23);
        else scale = __builtin_bit_cast(double, static_cast<uint64_t>(1023 - k) << 52);
        return scale;
    };
    auto expand_piece = [&](int piece, char *dst) {
        const int first = piece * HP, start = max(min(first, p.P2 - HP), 0);
        uint64_t raw;
        if constexpr (ES == 8) {
            raw = *reinterpret_cast<const uint64_t *>(gp + start);
        } else {
            const Chunk<S, HP> h = load_chunk<S, HP>(gp + start);
            uint64_t v;
            __builtin_memcpy(&v, h.e, 8);
            raw = v >> (static_cast<unsigned>(min(first - start, HP - 1)) * (8u * ES));
        }
        Chunk<S, E> out;
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            typename raw_t<ES>::type bits = static_cast<typename raw_t<ES>::type>(ES == 8 ? raw : (raw >> (h * 8 * (ES == 8 ? 0 : ES))));
            const S v = narrow<T>(widen<T>(__builtin_bit_cast(S, bits)) * pooled_scale(first + h));
            if constexpr (E >= 2) {
                out.e[2 * h] = v;
                out.e[2 * h + 1] = v;
            } else {
                out.e[0] = v;
            }
        }
        __builtin_memcpy(__builtin_assume_aligned(dst, 16), out.e, 16);
    };
    if (tid < xn) dma(reinterpret_cast<const char *>(xp), xlo + tid, 0);
    if constexpr (POOL) {
        if (tid < on) expand_piece(olo + tid, tile + SLOT + tid * 16);
        if (tid < sn) expand_piece(slo + tid, tile + 2 * SLOT + tid * 16);
    } else {
        if (tid < on) dma(gp16, olo + tid, SLOT);
        if (tid < sn) dma(gp16, slo + tid, 2 * SLOT);
    }
    if (wave == 0) {   // the pieces beyond 256 of each span: a handful of lanes
        if (tid + kThreads < xn) dma(reinterpret_cast<const char *>(xp), xlo + kThreads + tid, kThreads * 16);
        if constexpr (POOL) {
            if (tid + kThreads < on) expand_piece(olo + kThreads + tid, tile + SLOT + (kThreads + tid) * 16);
            if (tid + kThreads < sn) expand_piece(slo + kThreads + tid, tile + 2 * SLOT + (kThreads + tid) * 16);
        } else {
            if (tid + kThreads < on) dma(gp16, olo + kThreads + tid, SLOT + kThreads * 16);
            if (tid + kThreads < sn) dma(gp16, slo + kThreads + tid, 2 * SLOT + kThreads * 16);
        }
    }
    const int jc = sg * kThreads + tid, ji = jc * E;
    const bool mine = jc < cpr;
    ColState<E> xm, gm;
    {
        auto affine_state = [&](int first, int len, int safe) {
            ColState<E> st;
            st.base = first;
            if (first + E < 0 || first >= len) st.base = safe;
            st.affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) st.cm[e] = (first + e >= 0 && first + e < len) ? first + e : -1;
            return st;
        };
        if constexpr (PAD == 0) {
            xm = affine_state(ji - d.cx2, S2, xa0);
            gm = affine_state(ji - L2 - d.cg2, O2, sa0);
            if (O2 == 1) {   // a window one column wide ignores the shift: both corners read column 0
                gm.affine = false;
#pragma unroll
                for (int e = 0; e <= E; ++e) gm.cm[e] = (ji - L2 + e >= 0 && ji - L2 + e <= 1) ? 0 : -1;
            }
        } else {
            const size_t rec = (static_cast<size_t>(c) * cpr + (mine ? jc : 0)) * REC;
            xm = load_colstate<E>(p.colx + rec);
            gm = load_colstate<E>(p.colg + rec);
        }
    }
    auto in_span = [&](const ColState<E> &st, int c0, int c1) {
        bool ok = st.affine;
#pragma unroll
        for (int e = 0; e <= E; ++e) ok = ok && (st.cm[e] < 0 || (st.cm[e] >= c0 && st.cm[e] < c1));
        return ok;
    };
    const bool xs = (PAD == 0 && O2 != 1) || in_span(xm, xa0, xa1), gs = (PAD == 0 && O2 != 1) || in_span(gm, sa0, sa1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CT part = CT(0);
    if (mine) {
        const CT dw[1] = {static_cast<CT>(d.dw[0])};
        const S zero = static_cast<S>(0.0f);
        bool inside[E];
#pragma unroll
        for (int e = 0; e < E; ++e) inside[e] = static_cast<unsigned>(ji + e - L2) < static_cast<unsigned>(O2);
        auto read = [&](const char *lds_col0, const S *mem_col0, bool staged, const ColState<E> &st, S (&raw)[E + 1]) {
            if (PAD == 0 && O2 != 1) {   // (launch-uniform: zeros padding, every state affine and staged -- no data-dependent branch)
                lds_read_row_affine<S, E>(lds_col0, true, st, raw);
            } else if (staged) {
                lds_read_row<S, E>(lds_col0, true, st, raw);
            } else if (POOL && mem_col0 == gp) {   // (a gradient chunk that wraps / reflects beyond the staged span: expanded from memory)
#pragma unroll
                for (int e = 0; e <= E; ++e) {
                    const int col = st.cm[e] >= 0 ? st.cm[e] : 0;
                    raw[e] = st.cm[e] >= 0 ? narrow<T>(widen<T>(gp[col >> 1]) * pooled_scale(col >> 1)) : zero;
                }
            } else {
#pragma unroll
                for (int e = 0; e <= E; ++e) raw[e] = st.cm[e] >= 0 ? mem_col0[st.cm[e]] : zero;
            }
        };
        S xr[E + 1], gr[E + 1];
        read(tile - xlo * 16, xp, xs, xm, xr);
        read(tile + 2 * SLOT - slo * 16 + gph_s, gp, gs, gm, gr);
        const S *own = reinterpret_cast<const S *>(tile + SLOT - olo * 16 + gph_s);   // column 0 of the own-gradient span
        Chunk<S, E> res;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const S gl = own[clampi(ji - L2 + e, oa0, max(oa0, oa1 - 1))];   // (clamped into the staged columns, then masked)
            const S g = inside[e] ? gl : zero;
            part = fma_ct(widen<T>(g), widen<T>(xr[e + 1]) - widen<T>(xr[e]), part);
            if constexpr (ACTIVE) {
                const CT v[2] = {widen<T>(gr[e]), widen<T>(gr[e + 1])};
                res.e[e] = inside[e] ? narrow<T>(interp_t<T, 1>(v, dw)) : zero;
            } else {
                res.e[e] = inside[e] ? gr[e] : zero;
            }
        }
        store_chunk<S, E>(gxp + ji, res);
    }
    double *scratch = reinterpret_cast<double *>(tile + 3 * SLOT + 64);
    {
        const CT t = wave_total(part);
        if ((tid & 63) == 63) scratch[wave] = static_cast<double>(t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) acc += scratch[w];
        p.partials[bid] = acc;
    }
}

struct SpanFwdPlan {
    int ocp, cps, spp, P, wholeP;
    uint64_t total;
    size_t lds;
    bool ok;
};

SpanFwdPlan span_forward_plan(const Geometry &g, int es) {
    SpanFwdPlan s{};
    const int E = 16 / es;
    const int64_t oe = g.O[1] * g.O[2];
    s.ocp = static_cast<int>((oe * es + 15) / 16);   // (1-D: the last chunk of a ragged output row is partial; 2-D: whole pieces)
    // source rows a step can touch: the output rows of 256 chunks (+ the corner row)
    const int64_t rows = std::min<int64_t>(g.O[1], (static_cast<int64_t>(kThreads) * E + g.O[2] - 2) / g.O[2] + 1) + (g.active && g.nd == 2 ? 1 : 0);
    const int64_t cprx = (g.S[2] * es + 15) / 16 + 1;   // pieces that cover a (ragged) source row
    const int64_t budget = kSpanRounds * kThreads;
    // whole source rows when a step spans whole output rows (an output row of at most 256 chunks); rows longer than a step
    // stage only the columns the step's chunks reach
    if (g.O[2] * es <= kThreads * 16 && rows * cprx <= budget) {
        s.P = s.wholeP = static_cast<int>(cprx);
        s.cps = kThreads;
    } else {
        // only the columns the step's chunks reach: 256 chunks' source columns (+ the corner column, + the misalignment of the
        // shift) lie in 258 pieces.  (Steps of 254 chunks, whose columns fit one staging round of 256 pieces, were measured: their
        // 4064-byte output blocks split every 64-byte sector between two workgroups -- N256 C512 L4096 fp32 0.91 -> 1.10 ms.)
        s.wholeP = 0;
        s.cps = kThreads;
        s.P = kThreads + 2;
    }
    s.spp = (s.ocp + s.cps - 1) / s.cps;
    s.total = static_cast<uint64_t>(g.N) * g.C * s.spp;
    const int64_t slots = s.cps > kThreads ? 1 : rows;   // (a whole 1-D row per step: one source row)
    s.ok = slots * s.P <= (s.cps > kThreads ? 16 : kSpanRounds) * kThreads;
    s.lds = 64 + ((static_cast<size_t>(slots) * s.P * 16 + 63) & ~static_cast<size_t>(63)) + 64;
    return s;
}

struct SpanPlan {
    int cpr, seg, nseg, R, U, rsteps, spp, P, ndiff, rec;
    uint64_t total;
    size_t off_desc, off_colx, off_colg, bytes, lds;
};

SpanPlan span_plan(const Geometry &g, int es) {
    SpanPlan s{};
    const int E = 16 / es;
    const bool xrag = (g.S[2] * es) % 16 != 0;   // crop_backward<.., XRAG>: row-relative chunks, the last one partial
    s.cpr = static_cast<int>((g.S[2] * es + 15) / 16);
    if (s.cpr < 1) s.cpr = 1;
    // column segments of at most 256 chunks (4 KB blocks of grad_x: see span_forward_plan); slots of seg + 2 pieces (a ragged
    // grad_out row's cover; a segment's source columns + corner column + shift misalignment)
    s.seg = std::min(s.cpr, kThreads);
    s.nseg = (s.cpr + s.seg - 1) / s.seg;
    s.P = s.seg + 2;
    // rows per step: R * seg threads, and the tile's (3 R + 2) slots of P pieces within the staging rounds
    int R = std::max(1, g.nd >= 2 ? kThreads / (s.seg + 2) : kThreads / s.seg);   // (2-D / 3-D: R covers of cpr + 2 pieces per staging pass)
    R = std::max(1, std::min<int>(R, static_cast<int>(g.S[1])));
    if (g.nd == 1 || s.nseg > 1) R = 1;
    s.R = R;
    // row groups per thread (crop_backward<.., U>): two for the interpolating shift on rows of whole pieces (round 6: N64 C256 224x224 cut
    // 1/1 fp32 1.75 -> 1.615 ms, the sparse crop's 1.60; N512 C16 64x64 0.075 -> 0.069) and for the sparse shift on small tensors
    // (N512 C16 64x64: 0.068 -> 0.064 ms; N64 C256 224x224: 1.598 -> 1.616, so not there).  4- and 2-byte elements (fp64 keeps one).
    // Knob 35 bit 7: one for the sparse shift everywhere, bit 8: two everywhere.  Geometry and knobs only: the workspace is planned from the
    // same answer.
    s.U = 1;
    if (g.nd == 3 && g.active && g.K[0] <= 0 && es == 4) s.U = 2;   // (crop_backward3: the interpolating, unpooled form of 4-byte elements)
    if (g.nd == 2 && s.nseg == 1 && es <= 4) {
        // (ragged x rows, N512 C16 62x62 sparse: 0.072 -> 0.066 ms, N64 C256 222x222: 1.666 -> 1.668; the pooled interpolating form,
        //  N64 C256 224x224: 1.855 -> 1.668 ms -- the same rule for every form.  The interpolating shift has NO one-group instantiation
        //  for 4- / 2-byte elements: planes of fewer than 2 R rows leave the second group idle.)
        const int64_t steps1 = g.N * g.C * ((g.S[1] + R - 1) / R);
        if (g.active) s.U = 2;
        else if (!(g_step_tune[3] & 128) && g.S[1] >= 2 * R && (steps1 <= 65536 || (g_step_tune[3] & 256))) s.U = 2;
    }
    R *= s.U;   // rows per step
    s.rsteps = static_cast<int>((g.S[1] + R - 1) / R);
    s.spp = s.rsteps * s.nseg * (g.nd == 3 ? static_cast<int>(g.S[0]) : 1);   // (3-D: the steps of a whole (n, c) volume)
    s.ndiff = g.nd == 1 ? 1 : (g.nd == 2 ? 2 : 8);
    s.rec = (E + 3 <= 8) ? 8 : 16;
    s.total = static_cast<uint64_t>(g.N) * g.C * s.spp;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    s.off_desc = up(s.total * s.ndiff * sizeof(double));
    s.off_colx = s.off_desc + up(static_cast<size_t>(g.C) * sizeof(ChanDesc));
    s.off_colg = s.off_colx + up(static_cast<size_t>(g.C) * s.cpr * s.rec * sizeof(int16_t));
    s.bytes = s.off_colg + up(static_cast<size_t>(g.C) * s.cpr * s.rec * sizeof(int16_t));
    if (g.nd == 3) {   // crop_backward3: x rows [2][R + 1][cpr] | own rows [R][cpr + 2] | read rows [2][R + 1][cpr + 2]
        const size_t tile = (static_cast<size_t>(2 * (R + 1)) * s.cpr + static_cast<size_t>(R + 2 * (R + 1)) * (s.cpr + 2)) * 16;
        s.lds = 64 + ((tile + 63) & ~static_cast<size_t>(63)) + 64 + (kThreads / 64) * 8 * sizeof(double);
    } else if (g.nd == 2) {   // crop_backward: x rows [R + 1][cpr] | own rows [R][cpr + 2] | read rows [R + 1][cpr + 2]
        const size_t tile = (static_cast<size_t>(R + 1) * (s.cpr + (xrag ? 2 : 0)) + static_cast<size_t>(2 * R + 1) * (s.cpr + 2)) * 16;
        s.lds = 64 + ((tile + 63) & ~static_cast<size_t>(63)) + 64 + (kThreads / 64) * 2 * sizeof(double);
    } else {   // row_backward: three spans of (256 + 3) pieces
        s.lds = 64 + 3 * (kThreads + 3) * 16 + 64 + (kThreads / 64) * sizeof(double);
    }
    return s;
}

// what the kernel serves, pointers aside (the workspace is planned from this)
static bool span_geometry_ok(const Geometry &g, int dtype, bool pooled);
bool span_geometry_ok(const Geometry &g, int dtype) { return span_geometry_ok(g, dtype, false); }

// pooled: crop_backward<.., POOL> -- 2-D, x rows of whole pieces, 2 x 2 windows, pooled rows of at least half a piece (g.K / g.P set; the
// plan reads S, O, L only)
static bool span_geometry_ok(const Geometry &g, int dtype, bool pooled) {
    if (dtype > SHIFTND_BF16 || g.nd < 1 || g.nd > 3 || (g.K[0] > 0) != pooled) return false;
    const int es = dtype_size(dtype);
    // (pooled: 2 x 2 windows on 2-D planes; windows of 2 on 1-D rows -- row_backward<.., POOL>)
    if (pooled && ((g.S[2] * es) % 16 != 0 || g.K[0] != (g.nd == 3 ? 2 : 1) || g.K[1] != (g.nd >= 2 ? 2 : 1) || g.K[2] != 2 ||
                   g.P[2] < std::max(1, 8 / es) || (g.nd == 3 && es == 8))) return false;   // (3-D pooled: 2- / 4-byte elements)
    if (g.nd == 3) {   // crop_backward3: x rows of whole pieces, every dim of the volume and of the window at least 2
        if (g.S[0] < 2 || g.S[1] < 2 || g.S[2] < 2 || g.O[0] < 2 || g.O[1] < 2 || g.O[2] < 2 || (g.S[2] * es) % 16 != 0) return false;
        if (g.S[0] * g.S[1] * g.S[2] >= (1LL << 28) || g.O[0] * g.O[1] * g.O[2] >= (1LL << 28)) return false;   // 32-bit byte offsets within a volume
    } else if (g.S[0] != 1 || g.O[0] != 1) {
        return false;
    }
    if (g.S[1] < 1 || g.S[2] < 1 || g.O[1] < 1 || g.O[2] < 1) return false;
    // x rows: whole pieces -- or, 2-D with 4- / 8-byte elements, any length (crop_backward<.., XRAG>, round 5); int16 column tables
    // (2-byte elements: input AND gradient rows of an even number of elements -- every row at a 4-byte boundary)
    const bool rag_ok = g.nd == 2 && (es >= 4 || (es == 2 && g.S[2] % 2 == 0 && g.O[2] % 2 == 0));
    if (((g.S[2] * es) % 16 != 0 && !rag_ok) || g.S[2] > 32000) return false;
    // (grad_out / x need not be a whole number of pieces: the last piece of a cover reaches at most 15 bytes past the tensor's
    //  end, inside the 16-byte granule -- hence the page -- of its last valid byte; those bytes are never used)
    if (g.S[1] * g.S[2] >= (1LL << 28) || g.O[1] * g.O[2] >= (1LL << 28)) return false;  // 32-bit byte offsets within a plane
    if (g.nd >= 2 && g.S[2] * es > (kThreads - 2) * 16) return false;   // crop_backward: a grad_out row's cover (cpr + 2 pieces) per staging pass
    const SpanPlan s = span_plan(g, es);
    return s.total + 8 < (1ull << 31) && s.lds <= 64 * 1024;
}

}  // namespace

// the forward of cropped windows, 1-D rows and ragged rows: dense float tensors whose planes (output) and total size (input)
// are whole 16-byte pieces
static bool crop_forward_ok(const Geometry &g, int es);

// ragged_forward: 2-D, 4- / 8-byte elements, source rows that are not whole pieces, at least 8 chunks wide (shorter rows: whole planes
// through the flat-stream kernels) and at most 254 (a cover of xcpr + 2 pieces per staging pass), at least 16 rows
static bool ragged_forward_ok(const Geometry &g, int es) {
    // (knob 34 = 3: windows on source rows of whole pieces too -- crop_forward's share -- for A / B runs)
    const bool window = g.O[1] != g.S[1] || g.O[2] != g.S[2];
    if (g.nd != 2 || es < 2 || ((g.S[2] * es) % 16 == 0 && !(g_step_tune[2] == 3 && window))) return false;
    if (es == 2 && (g.S[2] % 2 != 0 || g.O[2] % 2 != 0)) return false;   // 2-byte elements: every source and output row at a 4-byte boundary
    if (g.S[2] * es < 8 * 16 || g.S[2] * es > (kThreads - 2) * 16 || g.S[1] < 16) return false;
    return g.S[1] * g.S[2] < (1LL << 28) && g.O[1] * g.O[2] < (1LL << 28);
}

// crop_forward3: cropped 3-D volumes, source rows of whole pieces (at most 256), every dim of the volume and of the window at least
// 2; 16-bit elements: output rows of an even number of elements (knob 35 bit 10 keeps the per-channel kernels)
static bool crop_forward3_ok(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g.nd != 3 || dtype > SHIFTND_BF16 || g.K[0] > 0 || (g_step_tune[3] & 1024)) return false;
    const int es = dtype_size(dtype);
    bool crop3 = false;
    for (int d = 0; d < 3; ++d) {
        if (g.S[d] < 2 || g.O[d] < 2) return false;
        crop3 = crop3 || g.O[d] != g.S[d] || g.L[d] != 0;
    }
    if (!crop3 || (g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads || (es == 2 && g.O[2] % 2 != 0)) return false;
    if (g.S[0] * g.S[1] * g.S[2] >= (1LL << 28) || g.O[0] * g.O[1] * g.O[2] >= (1LL << 28)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % (es < 4 ? 4 : es)) return false;
    const int64_t xcpr = g.S[2] * es / 16, rows = std::max<int64_t>(1, std::min<int64_t>(g.O[1], kThreads / xcpr));
    return g.N * g.C * g.O[0] * ((g.O[1] + rows - 1) / rows) + 8 < (1LL << 31);
}

// crop_forward3<.., ND = 2>: cropped 2-D windows whose output planes are not whole 16-byte pieces (crop_forward needs them to be), 2- and
// 4-byte elements, source rows of whole pieces (at most 256), rows and columns of at least 2; 16-bit: output rows of an even number of
// elements
static bool crop_rows_forward_ok(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g.nd != 2 || dtype > SHIFTND_BF16 || dtype == SHIFTND_F64 || g.K[0] > 0 || (g_step_tune[3] & 1024)) return false;
    const int es = dtype_size(dtype);
    if (g.S[0] != 1 || g.O[0] != 1 || g.S[1] < 2 || g.S[2] < 2 || g.O[1] < 2 || g.O[2] < 2) return false;
    bool crop = false;
    for (int d = 1; d < 3; ++d) crop = crop || g.O[d] != g.S[d] || g.L[d] != 0;
    if (!crop || (g.O[1] * g.O[2] * es) % 16 == 0) return false;   // (whole-piece planes: crop_forward)
    if ((g.S[2] * es) % 16 != 0 || g.S[2] * es / 16 > kThreads || (es == 2 && g.O[2] % 2 != 0)) return false;
    if (g.S[1] * g.S[2] >= (1LL << 28)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 4) return false;
    const int64_t xcpr = g.S[2] * es / 16, rows = std::max<int64_t>(1, std::min<int64_t>(g.O[1], kThreads / xcpr));
    // (a step is `rows` output rows of ONE plane: planes too small to fill half a workgroup keep the flat-stream kernels, which pack
    //  many planes into a step)
    if (rows * ((g.O[2] * es + 15) / 16) < kThreads / 2) return false;
    return g.N * g.C * ((g.O[1] + rows - 1) / rows) + 8 < (1LL << 31);
}

bool span_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[2] == 1) return false;   // knob 34 = 1: no forwards through LDS
    if (g.nd == 3) return crop_forward3_ok(g, dtype, x, out);
    if (crop_rows_forward_ok(g, dtype, x, out)) return true;
    if (dtype > SHIFTND_BF16 || (g.nd != 1 && g.nd != 2) || g.K[0] > 0) return false;
    const int es = dtype_size(dtype);
    if (g.S[0] != 1 || g.O[0] != 1 || g.S[1] < 1 || g.S[2] < 1 || g.O[1] < 1 || g.O[2] < 1) return false;
    if (ragged_forward_ok(g, es)) {
        if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
        if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % es) return false;
        const int64_t rows = std::min<int64_t>(g.O[1], kThreads / ((g.S[2] * es + 15) / 16 + 2));
        return g.N * g.C * ((g.O[1] + rows - 1) / rows) + 8 < (1LL << 31);
    }
    if (g.nd == 1 && (g.S[2] * es) % 16 != 0) return false;   // row_forward: SOURCE rows of whole pieces
    // (1-D output rows of any length at a 4-byte boundary -- round 6; 2-D: output planes of whole pieces)
    const bool ragged_row1 = g.nd == 1 && (g.O[2] * es) % 16 != 0 && (es >= 4 || g.O[2] % 2 == 0);
    if (((g.O[1] * g.O[2] * es) % 16 != 0 && !ragged_row1) || (g.N * g.C * g.S[1] * g.S[2] * es) % 16 != 0) return false;
    if (g.S[1] * g.S[2] >= (1LL << 28) || g.O[1] * g.O[2] >= (1LL << 28)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16) return false;
    const SpanFwdPlan s = span_forward_plan(g, es);
    if (!s.ok || s.total + 8 >= (1ull << 31) || s.lds > 64 * 1024) return false;
    const bool served = g.nd == 1 ? (g.S[2] * es) % 16 == 0 : crop_forward_ok(g, es);   // row_forward / crop_forward
    if (!served) return false;
    if (g_step_tune[2] >= 2) return true;
    bool crop = false;
    for (int d = 1; d < 3; ++d) crop = crop || g.O[d] != g.S[d] || g.L[d] != 0;
    // cropped 2-D windows on source rows of whole pieces (the aligned ones whose output rows are whole pieces too: the step kernels,
    // asked first) and 1-D rows of at least 128 chunks (same box, N256 C512 L4096: fp32 0.94 -> 0.69 ms, interpolating 0.84 -> 0.69,
    // fp16 0.44 -> 0.35).  Ragged source rows: shiftnd_flat.hip.
    if (g.nd == 1) return g.O[2] * es / 16 >= 128;   // row_forward (short rows: the per-channel kernels)
    return crop;
}

// crop_forward: 2-D, source rows of whole pieces (at most 256), at most four staging rounds
// source rows a step of `chunks` output chunks stages (the corner row of the interpolating shift included)
static int64_t crop_forward_rows(const Geometry &g, int es, int chunks) {
    const int E = 16 / es;
    return std::min<int64_t>(g.O[1], (static_cast<int64_t>(chunks) * E + g.O[2] - 2) / g.O[2] + 1) + (g.active ? 1 : 0);
}

static bool crop_forward_ok(const Geometry &g, int es) {
    if (g.nd != 2 || (g.S[2] * es) % 16 != 0 || g.S[2] * es > kThreads * 16) return false;
    return crop_forward_rows(g, es, kThreads) * (g.S[2] * es / 16) <= 4 * kThreads;
}

// chunks per thread of crop_forward: two for the interpolating shift on planes of more than one such step whose rows fit 24 KiB of LDS
// (knob 35 bit 1 = 2: always one)
static int crop_forward_groups(const Geometry &g, int es) {
    if (!g.active || (g_step_tune[3] & 2)) return 1;
    const int64_t ocp = (g.O[1] * g.O[2] * es + 15) / 16;
    if (ocp <= 2 * kThreads) return 1;
    return crop_forward_rows(g, es, 2 * kThreads) * (g.S[2] * es / 16) <= 6 * kThreads ? 2 : 1;
}

// Shift3d + avg_pool3d(2) in one pass (crop_forward3_pool, round 6): 2 x 2 x 2 windows, cropped or not, both shifts; 2- / 4-byte float
// elements, source rows of whole pieces (at most 85: 2 R + 1 staged rows per plane with R >= 1), every dim of the volume and of the
// window at least 2
bool span_forward_pooled3_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[2] == 1 || (g_step_tune[3] & 1024) || g.nd != 3 || g.K[0] != 2 || g.K[1] != 2 || g.K[2] != 2) return false;
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)
        if (g.S[d] < 2 || g.O[d] < 2) return false;
    if ((g.S[2] * es) % 16 != 0 || 3 * (g.S[2] * es / 16) > kThreads) return false;
    if (g.S[0] * g.S[1] * g.S[2] >= (1LL << 28)) return false;
    if (!dense(g.xs, g.N, g.C, g.S)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % (es < 4 ? 4 : es)) return false;
    const int64_t xcpr = g.S[2] * es / 16, R = std::max<int64_t>(1, std::min<int64_t>(g.P[1], (kThreads / xcpr - 1) / 2));
    return g.N * g.C * g.P[0] * ((g.P[1] + R - 1) / R) + 8 < (1LL << 31);
}

int span_forward_pooled3(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    SpanFwdParams p{};
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
    p.P0 = static_cast<int>(g.P[0]);
    p.P1 = static_cast<int>(g.P[1]);
    p.P2 = static_cast<int>(g.P[2]);
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.o_plane = g.P[0] * g.P[1] * g.P[2];
    p.P = static_cast<int>(g.S[2] * es / 16);                      // pieces per staged source row
    p.ocp = static_cast<int>((g.O[2] * es + 15) / 16);             // chunks per virtual output row
    p.cps = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(g.P[1], (kThreads / p.P - 1) / 2)));   // pooled rows per step
    p.rsteps = (p.P1 + p.cps - 1) / p.cps;
    p.spp = p.rsteps * p.P0;
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_O2 = make_fastdiv(static_cast<uint32_t>(p.ocp));
    p.d_P = make_fastdiv(static_cast<uint32_t>(p.P));
    p.d_rsteps = make_fastdiv(static_cast<uint32_t>(p.rsteps));
    p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const bool act = g.active != 0;
    const size_t lds = 64 + static_cast<size_t>(act ? 3 : 2) * (2 * p.cps + 1) * p.P * 16 + 64;
    note_kernel("crop_forward3_pool");
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_CROP3_POOL(TT, ACT) \
    switch (pad_template(g.pad)) { \
    case 0: hipLaunchKernelGGL((crop_forward3_pool<TT, ACT, 0>), grid, block, lds, st, p); break; \
    case 1: hipLaunchKernelGGL((crop_forward3_pool<TT, ACT, 1>), grid, block, lds, st, p); break; \
    case 2: hipLaunchKernelGGL((crop_forward3_pool<TT, ACT, 2>), grid, block, lds, st, p); break; \
    default: hipLaunchKernelGGL((crop_forward3_pool<TT, ACT, kPadMirror>), grid, block, lds, st, p); break; \
    }
#define SHIFTND_CROP3_POOL_T(TT) \
    if (act) { SHIFTND_CROP3_POOL(TT, true) } else { SHIFTND_CROP3_POOL(TT, false) }
    if (dtype == SHIFTND_F32) { SHIFTND_CROP3_POOL_T(f32_t) } else if (dtype == SHIFTND_F16) { SHIFTND_CROP3_POOL_T(f16_t) } else { SHIFTND_CROP3_POOL_T(bf16_t) }
#undef SHIFTND_CROP3_POOL_T
#undef SHIFTND_CROP3_POOL
    return SHIFTND_OK;
}

// Shift1d + avg_pool1d(2) in one pass (row_forward<.., POOL>, round 6): 2- / 4-byte float elements, source rows of whole pieces, at
// least 128 chunks (shorter rows: the per-channel kernels, like the unpooled rule)
bool span_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (g_step_tune[2] == 1 || g.nd != 1 || g.K[2] != 2 || g.K[1] > 1) return false;
    if (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16) return false;
    const int es = dtype_size(dtype);
    if (g.S[0] != 1 || g.S[1] != 1 || g.O[0] != 1 || g.O[1] != 1 || g.S[2] < 2 || g.O[2] < 1) return false;
    if ((g.S[2] * es) % 16 != 0 || g.S[2] >= (1LL << 28) || (g.N * g.C * g.S[2] * es) % 16 != 0) return false;
    if (!dense(g.xs, g.N, g.C, g.S)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % (es < 4 ? 4 : es)) return false;
    const int64_t ocp = (g.O[2] * es + 15) / 16;
    if (g.N * g.C * ((ocp + kThreads - 1) / kThreads) + 8 >= (1LL << 31)) return false;
    return g_step_tune[2] >= 2 || ocp >= 128;
}

int span_forward_pooled(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    SpanFwdParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.nd = 1;
    p.S1 = p.O1 = 1;
    p.S2 = static_cast<int>(g.S[2]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L2 = static_cast<int>(g.L[2]);
    p.P2 = static_cast<int>(g.P[2]);
    p.x_plane = g.S[2];
    p.o_plane = g.P[2];   // (the pooled row)
    p.ocp = static_cast<int>((g.O[2] * es + 15) / 16);
    p.cps = kThreads;
    p.spp = (p.ocp + kThreads - 1) / kThreads;
    p.P = kThreads + 2;
    const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
    p.total_steps = static_cast<uint32_t>(total);
    p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_O2 = make_fastdiv(static_cast<uint32_t>(p.O2));
    p.d_P = make_fastdiv(static_cast<uint32_t>(p.P));
    p.pad = g.pad;
    p.d_per1 = make_fastdiv(1u);
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const size_t lds = 64 + (kThreads + 3) * 16 + 64;
    note_kernel("row_forward_pool");
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_ROW_POOL(TT, ACT) \
    switch (pad_template(g.pad)) { \
    case 0: hipLaunchKernelGGL((row_forward<TT, ACT, 0, true>), grid, block, lds, st, p); break; \
    case 1: hipLaunchKernelGGL((row_forward<TT, ACT, 1, true>), grid, block, lds, st, p); break; \
    case 2: hipLaunchKernelGGL((row_forward<TT, ACT, 2, true>), grid, block, lds, st, p); break; \
    default: hipLaunchKernelGGL((row_forward<TT, ACT, kPadMirror, true>), grid, block, lds, st, p); break; \
    }
#define SHIFTND_ROW_POOL_T(TT) \
    if (g.active) { SHIFTND_ROW_POOL(TT, true) } else { SHIFTND_ROW_POOL(TT, false) }
    if (dtype == SHIFTND_F32) { SHIFTND_ROW_POOL_T(f32_t) } else if (dtype == SHIFTND_F16) { SHIFTND_ROW_POOL_T(f16_t) } else { SHIFTND_ROW_POOL_T(bf16_t) }
#undef SHIFTND_ROW_POOL_T
#undef SHIFTND_ROW_POOL
    return SHIFTND_OK;
}

int span_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const bool rows2 = g.nd == 2 && crop_rows_forward_ok(g, dtype, x, out);
    if (g.nd == 3 || rows2) {   // crop_forward3 (3-D volumes; 2-D windows whose planes are not whole pieces)
        SpanFwdParams p{};
        p.x = x;
        p.out = out;
        p.w = w;
        p.wkind = wkind;
        p.C = static_cast<int>(g.C);
        p.nd = g.nd;
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
        p.P = static_cast<int>(g.S[2] * es / 16);                      // pieces per staged source row
        p.ocp = static_cast<int>((g.O[2] * es + 15) / 16);             // output chunks per row
        p.cps = std::max(1, std::min<int>(static_cast<int>(g.O[1]), kThreads / p.P));   // rows per row group
        const int kU = (g.active && (es == 4 || g.nd == 2)) ? 2 : 1;                    // row groups per thread (crop_forward3: U)
        p.rsteps = (p.O1 + kU * p.cps - 1) / (kU * p.cps);
        p.spp = p.rsteps * p.O0;
        const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
        p.total_steps = static_cast<uint32_t>(total);
        p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
        p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
        p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
        p.d_O2 = make_fastdiv(static_cast<uint32_t>(p.ocp));           // (thread -> (row, chunk))
        p.d_P = make_fastdiv(static_cast<uint32_t>(p.P));
        p.d_rsteps = make_fastdiv(static_cast<uint32_t>(p.rsteps));
        p.d_per0 = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
        p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
        p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
        const bool act = g.active != 0;
        const size_t lds = 64 + static_cast<size_t>(act ? 2 : 1) * (kU * p.cps + 1) * p.P * 16 + 64;
        if (rows2) note_kernel(act ? "crop_active_forward_rows" : "crop_gather_forward_rows");
        else note_kernel(act ? "crop_active_forward3" : "crop_gather_forward3");
        const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_CROP3_FWD_ND(TT, ACT, NDV) \
        switch (pad_template(g.pad)) { \
        case 0: hipLaunchKernelGGL((crop_forward3<TT, ACT, 0, NDV>), grid, block, lds, st, p); break; \
        case 1: hipLaunchKernelGGL((crop_forward3<TT, ACT, 1, NDV>), grid, block, lds, st, p); break; \
        case 2: hipLaunchKernelGGL((crop_forward3<TT, ACT, 2, NDV>), grid, block, lds, st, p); break; \
        default: hipLaunchKernelGGL((crop_forward3<TT, ACT, kPadMirror, NDV>), grid, block, lds, st, p); break; \
        }
        if (rows2) {   // 2- and 4-byte elements
            if (!act) {
                if (es == 2) { SHIFTND_CROP3_FWD_ND(f16_t, false, 2) } else { SHIFTND_CROP3_FWD_ND(f32_t, false, 2) }
            } else if (dtype == SHIFTND_F32) { SHIFTND_CROP3_FWD_ND(f32_t, true, 2)
            } else if (dtype == SHIFTND_F16) { SHIFTND_CROP3_FWD_ND(f16_t, true, 2)
            } else { SHIFTND_CROP3_FWD_ND(bf16_t, true, 2) }
            return SHIFTND_OK;
        }
#define SHIFTND_CROP3_FWD(TT, ACT) SHIFTND_CROP3_FWD_ND(TT, ACT, 3)
        if (!act) {   // a raw copy (the weights are widened by their own dtype, p.wkind): one instantiation per element size
            if (es == 2) { SHIFTND_CROP3_FWD(f16_t, false) } else if (es == 4) { SHIFTND_CROP3_FWD(f32_t, false) } else { SHIFTND_CROP3_FWD(f64_t, false) }
        } else if (dtype == SHIFTND_F32) { SHIFTND_CROP3_FWD(f32_t, true)
        } else if (dtype == SHIFTND_F64) { SHIFTND_CROP3_FWD(f64_t, true)
        } else if (dtype == SHIFTND_F16) { SHIFTND_CROP3_FWD(f16_t, true)
        } else { SHIFTND_CROP3_FWD(bf16_t, true) }
#undef SHIFTND_CROP3_FWD
#undef SHIFTND_CROP3_FWD_ND
        return SHIFTND_OK;
    }
    if (ragged_forward_ok(g, es)) {
        SpanFwdParams p{};
        p.x = x;
        p.out = out;
        p.w = w;
        p.wkind = wkind;
        p.C = static_cast<int>(g.C);
        p.nd = g.nd;
        p.S1 = static_cast<int>(g.S[1]);
        p.S2 = static_cast<int>(g.S[2]);
        p.O1 = static_cast<int>(g.O[1]);
        p.O2 = static_cast<int>(g.O[2]);
        p.L1 = static_cast<int>(g.L[1]);
        p.L2 = static_cast<int>(g.L[2]);
        p.x_plane = g.S[1] * g.S[2];
        p.o_plane = g.O[1] * g.O[2];
        const int xcpr = static_cast<int>((g.S[2] * es + 15) / 16);
        p.P = xcpr + 2;                                                 // pieces per staged source row
        p.ocp = static_cast<int>((g.O[2] * es + 15) / 16);              // output chunks per row
        p.cps = std::max(1, std::min<int>(static_cast<int>(g.O[1]), kThreads / p.P));   // rows per step
        p.spp = (p.O1 + p.cps - 1) / p.cps;
        const uint64_t total = static_cast<uint64_t>(g.N) * g.C * p.spp;
        p.total_steps = static_cast<uint32_t>(total);
        p.steps_per_xcd = static_cast<uint32_t>((total + 7) / 8);
        p.d_spp = make_fastdiv(static_cast<uint32_t>(p.spp));
        p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
        p.d_O2 = make_fastdiv(static_cast<uint32_t>(p.ocp));            // (ragged_forward: thread -> (row, chunk))
        p.d_P = make_fastdiv(static_cast<uint32_t>(p.P));
        p.pad = g.pad;
        p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
        p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
        const size_t lds = 64 + static_cast<size_t>(p.cps + 1) * p.P * 16 + 64;
        const bool act = g.active != 0;
        note_kernel(act ? "ragged_active_forward" : "ragged_gather_forward");
        const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_RAG_FWD(TT, ACT) \
        switch (pad_template(g.pad)) { \
        case 0: hipLaunchKernelGGL((ragged_forward<TT, ACT, 0>), grid, block, lds, st, p); break; \
        case 1: hipLaunchKernelGGL((ragged_forward<TT, ACT, 1>), grid, block, lds, st, p); break; \
        case 2: hipLaunchKernelGGL((ragged_forward<TT, ACT, 2>), grid, block, lds, st, p); break; \
        default: hipLaunchKernelGGL((ragged_forward<TT, ACT, kPadMirror>), grid, block, lds, st, p); break; \
        }
        if (!act) {   // a raw copy: one instantiation per element size
            if (es == 2) { SHIFTND_RAG_FWD(f16_t, false) } else if (es == 4) { SHIFTND_RAG_FWD(f32_t, false) } else { SHIFTND_RAG_FWD(f64_t, false) }
        } else if (dtype == SHIFTND_F32) { SHIFTND_RAG_FWD(f32_t, true)
        } else if (dtype == SHIFTND_F64) { SHIFTND_RAG_FWD(f64_t, true)
        } else if (dtype == SHIFTND_F16) { SHIFTND_RAG_FWD(f16_t, true)
        } else { SHIFTND_RAG_FWD(bf16_t, true) }
#undef SHIFTND_RAG_FWD
        return SHIFTND_OK;
    }
    SpanFwdPlan sp = span_forward_plan(g, es);
    const bool row1d = g.nd == 1 && (g.S[2] * es) % 16 == 0;
    if (row1d) {   // row_forward: segments of 256 chunks, the columns their windows reach
        sp.P = kThreads + 2;
        sp.wholeP = 0;
        sp.cps = kThreads;
        sp.spp = (sp.ocp + kThreads - 1) / kThreads;
        sp.total = static_cast<uint64_t>(g.N) * g.C * sp.spp;
        sp.lds = 64 + (kThreads + 3) * 16 + 64;
    }
    const bool lean = crop_forward_ok(g, es);
    int cropU = 1;
    if (lean) {   // whole source rows, exact pitch
        cropU = crop_forward_groups(g, es);
        const int64_t rows = crop_forward_rows(g, es, cropU * kThreads);
        sp.P = sp.wholeP = static_cast<int>(g.S[2] * es / 16);
        sp.cps = cropU * kThreads;
        sp.spp = (sp.ocp + sp.cps - 1) / sp.cps;
        sp.total = static_cast<uint64_t>(g.N) * g.C * sp.spp;
        sp.lds = 64 + ((static_cast<size_t>(rows) * sp.P * 16 + 63) & ~static_cast<size_t>(63)) + 64;
    }
    SpanFwdParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.x_plane = g.S[1] * g.S[2];
    p.o_plane = g.O[1] * g.O[2];
    p.ocp = sp.ocp;
    p.cps = sp.cps;
    p.spp = sp.spp;
    p.P = sp.P;
    p.wholeP = sp.wholeP;
    p.total_steps = static_cast<uint32_t>(sp.total);
    p.steps_per_xcd = static_cast<uint32_t>((sp.total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(sp.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_O2 = make_fastdiv(static_cast<uint32_t>(p.O2));
    p.d_P = make_fastdiv(static_cast<uint32_t>(sp.P));
    p.pad = g.pad;
    p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    const bool active = g.active != 0;
    if (row1d) {
        note_kernel(active ? "row_active_forward" : "row_gather_forward");
        const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_ROW_FWD(TT, ACT) \
        switch (pad_template(g.pad)) { \
        case 0: hipLaunchKernelGGL((row_forward<TT, ACT, 0>), grid, block, sp.lds, st, p); break; \
        case 1: hipLaunchKernelGGL((row_forward<TT, ACT, 1>), grid, block, sp.lds, st, p); break; \
        case 2: hipLaunchKernelGGL((row_forward<TT, ACT, 2>), grid, block, sp.lds, st, p); break; \
        default: hipLaunchKernelGGL((row_forward<TT, ACT, kPadMirror>), grid, block, sp.lds, st, p); break; \
        }
        if (!active) {
            if (es == 2) { SHIFTND_ROW_FWD(f16_t, false) } else if (es == 4) { SHIFTND_ROW_FWD(f32_t, false) } else { SHIFTND_ROW_FWD(f64_t, false) }
        } else if (dtype == SHIFTND_F32) { SHIFTND_ROW_FWD(f32_t, true)
        } else if (dtype == SHIFTND_F64) { SHIFTND_ROW_FWD(f64_t, true)
        } else if (dtype == SHIFTND_F16) { SHIFTND_ROW_FWD(f16_t, true)
        } else { SHIFTND_ROW_FWD(bf16_t, true) }
#undef SHIFTND_ROW_FWD
        return SHIFTND_OK;
    }
    if (lean) {
        note_kernel(active ? "crop_active_forward" : "crop_gather_forward");
        const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_CROP_FWD_U(TT, ACT, UU) \
        switch (pad_template(g.pad)) { \
        case 0: hipLaunchKernelGGL((crop_forward<TT, ACT, 0, UU>), grid, block, sp.lds, st, p); break; \
        case 1: hipLaunchKernelGGL((crop_forward<TT, ACT, 1, UU>), grid, block, sp.lds, st, p); break; \
        case 2: hipLaunchKernelGGL((crop_forward<TT, ACT, 2, UU>), grid, block, sp.lds, st, p); break; \
        default: hipLaunchKernelGGL((crop_forward<TT, ACT, kPadMirror, UU>), grid, block, sp.lds, st, p); break; \
        }
#define SHIFTND_CROP_FWD(TT, ACT) \
        if (ACT && cropU == 2) { SHIFTND_CROP_FWD_U(TT, ACT, (ACT ? 2 : 1)) } else { SHIFTND_CROP_FWD_U(TT, ACT, 1) }
        if (!active) {   // a raw copy: one instantiation per element size
            if (es == 2) { SHIFTND_CROP_FWD(f16_t, false) } else if (es == 4) { SHIFTND_CROP_FWD(f32_t, false) } else { SHIFTND_CROP_FWD(f64_t, false) }
        } else if (dtype == SHIFTND_F32) { SHIFTND_CROP_FWD(f32_t, true)
        } else if (dtype == SHIFTND_F64) { SHIFTND_CROP_FWD(f64_t, true)
        } else if (dtype == SHIFTND_F16) { SHIFTND_CROP_FWD(f16_t, true)
        } else { SHIFTND_CROP_FWD(bf16_t, true) }
#undef SHIFTND_CROP_FWD
#undef SHIFTND_CROP_FWD_U
        return SHIFTND_OK;
    }
    return SHIFTND_ERR_INVALID_ARGUMENT;   // (span_forward_eligible admits nothing else)
}

// cropped 2-D problems, and 1-D problems whose rows fill at least a wave: dense tensors, 16-byte aligned
bool span_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g_step_tune[0] == 1 || !span_geometry_ok(g, dtype)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.os, g.N, g.C, g.O) || !dense(g.gs, g.N, g.C, g.S)) return false;
    if (reinterpret_cast<uintptr_t>(go) % 16 || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(gx) % 16) return false;
    // (crop_backward<.., PAD = 0> reads every chunk through an affine column state; a window one column wide ignores the shift and
    //  is not one)
    if (g.nd == 2 && g.pad == 0 && g.O[2] == 1) return false;
    if (g.nd == 3) {   // crop_backward3: cropped volumes only (the walk kernels take the others); knob 35 bit 10 keeps the plane kernels
        bool crop3 = false;
        for (int d = 0; d < 3; ++d) crop3 = crop3 || g.O[d] != g.S[d] || g.L[d] != 0;
        return crop3 && !(g_step_tune[3] & 1024);
    }
    if (g_step_tune[0] == 2) return true;
    const int es = dtype_size(dtype);
    if (g.nd == 1) return g.S[2] * es / 16 >= 128;   // (short rows: one row per workgroup would leave most lanes idle)
    bool crop = false;
    for (int d = 1; d < 3; ++d) crop = crop || g.O[d] != g.S[d] || g.L[d] != 0;
    // ragged x rows of at least 8 chunks (62 x 62, 222 x 222 fp32 ...): the row-relative form; shorter rows (14 x 14, 7 x 7) leave most
    // of a workgroup's lanes idle here -- whole planes through the flat-stream kernels
    if ((g.S[2] * es) % 16 != 0) return g.S[2] * es >= 8 * 16 && g.S[1] >= 16;
    return crop;   // (uncropped 2-D: step_backward)
}

size_t span_backward_workspace(const Geometry &g, int dtype) { return span_geometry_ok(g, dtype) ? span_plan(g, dtype_size(dtype)).bytes : 0; }

// the fused shift + average-pool backward of 2-D problems (round 6): `go` = gradient of the pooled window, contiguous.  Cropped
// windows (every emulate_dw with padding < kernel / 2) and the interpolating shift, which step_backward<.., POOL> does not take
bool span_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g_step_tune[0] == 1 || !span_geometry_ok(g, dtype, true)) return false;
    if (!dense(g.xs, g.N, g.C, g.S) || !dense(g.gs, g.N, g.C, g.S)) return false;
    if (reinterpret_cast<uintptr_t>(go) % dtype_size(dtype) || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(gx) % 16) return false;
    if (g.pad == 0 && g.O[2] == 1) return false;   // (as span_backward_eligible: the affine column state)
    if (g.nd == 1) return g.S[2] * dtype_size(dtype) / 16 >= 128 || g_step_tune[0] == 2;   // (short rows: the per-channel kernels, as unpooled)
    if (g.nd == 3) {   // cropped volumes (the walk takes the others), as crop_backward3
        bool crop3 = false;
        for (int d = 0; d < 3; ++d) crop3 = crop3 || g.O[d] != g.S[d] || g.L[d] != 0;
        return crop3 && !(g_step_tune[3] & 1024);
    }
    return true;
}
size_t span_backward_pooled_workspace(const Geometry &g, int dtype) { return span_geometry_ok(g, dtype, true) ? span_plan(g, dtype_size(dtype)).bytes : 0; }

template <typename T, int ND, bool XRAG = false, bool POOL = false>
static void launch_span_backward(const SpanParams &p, const SpanPlan &sp, bool active, void *gw, hipStream_t st) {
    const dim3 grid(p.steps_per_xcd * 8), block(kThreads);
#define SHIFTND_SPAN_PAD(ACT, PADV) \
    case PADV: \
        if constexpr (ND == 2 && sizeof(typename T::S) <= 4 && ACT) { \
            hipLaunchKernelGGL((crop_backward<T, ACT, PADV, XRAG, POOL, 2>), grid, block, sp.lds, st, p); \
        } else if constexpr (ND == 2 && sizeof(typename T::S) <= 4) { \
            if (sp.U == 2) hipLaunchKernelGGL((crop_backward<T, ACT, PADV, XRAG, POOL, 2>), grid, block, sp.lds, st, p); \
            else hipLaunchKernelGGL((crop_backward<T, ACT, PADV, XRAG, POOL, 1>), grid, block, sp.lds, st, p); \
        } else if constexpr (ND == 2) { \
            hipLaunchKernelGGL((crop_backward<T, ACT, PADV, XRAG, POOL>), grid, block, sp.lds, st, p); \
        } else if constexpr (ND == 3) { \
            hipLaunchKernelGGL((crop_backward3<T, ACT, PADV, POOL>), grid, block, sp.lds, st, p); \
        } else { \
            hipLaunchKernelGGL((row_backward<T, ACT, PADV, POOL>), grid, block, sp.lds, st, p); \
        } \
        break;
    // (the channel descriptors come from span_prep for every padding: computing them in crop_backward itself -- tried in round 5 to save
    //  the 4.5 us launch -- put weight loads and 64-bit shift arithmetic in front of every one-step workgroup's first DMA: N64 C256
    //  224x224 cut 1/1 fp32 1.65 -> 2.02 ms)
    if (active) {
        hipLaunchKernelGGL((span_prep<T>), dim3(p.C), block, 0, st, p, true);
        switch (pad_template(p.pad)) { SHIFTND_SPAN_PAD(true, 0) SHIFTND_SPAN_PAD(true, 1) SHIFTND_SPAN_PAD(true, 2) default: SHIFTND_SPAN_PAD(true, 3) }
    } else {
        hipLaunchKernelGGL((span_prep<T>), dim3(p.C), block, 0, st, p, false);
        switch (pad_template(p.pad)) { SHIFTND_SPAN_PAD(false, 0) SHIFTND_SPAN_PAD(false, 1) SHIFTND_SPAN_PAD(false, 2) default: SHIFTND_SPAN_PAD(false, 3) }
    }
#undef SHIFTND_SPAN_PAD
    // the channel sums and the blends: step_reduce reads the record layout through StepParams
    StepParams r{};
    r.partials = p.partials;
    r.desc = p.desc;
    r.N = p.N;
    r.C = p.C;
    r.spv = p.spp;
    r.d_spv = p.d_spp;
    launch_step_reduce(T::kDtype, ND, r, gw, st);
}

int span_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                  hipStream_t st) {
    const int es = dtype_size(dtype);
    const SpanPlan sp = span_plan(g, es);
    SpanParams p{};
    p.x = x;
    p.go = go;
    p.out = gx;
    p.w = w;
    char *ws = static_cast<char *>(workspace);
    p.partials = reinterpret_cast<double *>(ws);
    p.desc = reinterpret_cast<ChanDesc *>(ws + sp.off_desc);
    p.colx = reinterpret_cast<int16_t *>(ws + sp.off_colx);
    p.colg = reinterpret_cast<int16_t *>(ws + sp.off_colg);
    p.x_plane = g.S[0] * g.S[1] * g.S[2];   // (2-D: S0 = O0 = 1)
    p.g_plane = g.O[0] * g.O[1] * g.O[2];
    p.S0 = static_cast<int>(g.S[0]);
    p.O0 = static_cast<int>(g.O[0]);
    p.L0 = static_cast<int>(g.L[0]);
    const bool pooled = g.K[0] > 0;
    if (pooled) {   // `go` is the gradient of the pooled window [P1, P2]
        p.P2 = static_cast<int>(g.P[2]);
        p.P1 = static_cast<int>(g.P[1]);
        p.g_plane = g.P[0] * g.P[1] * g.P[2];   // (2-D / 1-D: P0 = 1)
    }
    p.wkind = dtype;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.pad = g.pad;
    p.nd = g.nd;
    p.S1 = static_cast<int>(g.S[1]);
    p.S2 = static_cast<int>(g.S[2]);
    p.O1 = static_cast<int>(g.O[1]);
    p.O2 = static_cast<int>(g.O[2]);
    p.L1 = static_cast<int>(g.L[1]);
    p.L2 = static_cast<int>(g.L[2]);
    p.cpr = sp.cpr;
    p.seg = sp.seg;
    p.nseg = sp.nseg;
    p.R = sp.R;
    p.rsteps = sp.rsteps;
    p.spp = sp.spp;
    p.P = sp.P;
    p.total_steps = static_cast<uint32_t>(sp.total);
    p.steps_per_xcd = static_cast<uint32_t>((sp.total + 7) / 8);
    p.d_spp = make_fastdiv(static_cast<uint32_t>(sp.spp));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    p.d_seg = make_fastdiv(static_cast<uint32_t>(sp.seg));
    p.d_nseg = make_fastdiv(static_cast<uint32_t>(sp.nseg));
    p.d_P = make_fastdiv(static_cast<uint32_t>(sp.P));
    p.d_per1x = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
    p.d_per2x = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
    p.d_per1g = make_fastdiv(static_cast<uint32_t>(map_period(p.O1, g.pad)));
    p.d_per2g = make_fastdiv(static_cast<uint32_t>(map_period(p.O2, g.pad)));
    p.d_rsteps = make_fastdiv(static_cast<uint32_t>(sp.rsteps));
    p.d_per0x = make_fastdiv(static_cast<uint32_t>(map_period(p.S0, g.pad)));
    p.d_per0g = make_fastdiv(static_cast<uint32_t>(map_period(p.O0, g.pad)));
    const bool active = g.active != 0;
    if (g.nd == 3 && pooled) {
        note_kernel("crop_backward3_pool");
        switch (dtype) {
        case SHIFTND_F32: launch_span_backward<f32_t, 3, false, true>(p, sp, active, gw, st); break;
        case SHIFTND_F16: launch_span_backward<f16_t, 3, false, true>(p, sp, active, gw, st); break;
        default: launch_span_backward<bf16_t, 3, false, true>(p, sp, active, gw, st); break;
        }
        return SHIFTND_OK;
    }
    if (g.nd == 3) {
        note_kernel("crop_backward3");
        switch (dtype) {
        case SHIFTND_F64: launch_span_backward<f64_t, 3>(p, sp, active, gw, st); break;
        case SHIFTND_F32: launch_span_backward<f32_t, 3>(p, sp, active, gw, st); break;
        case SHIFTND_F16: launch_span_backward<f16_t, 3>(p, sp, active, gw, st); break;
        default: launch_span_backward<bf16_t, 3>(p, sp, active, gw, st); break;
        }
        return SHIFTND_OK;
    }
    if (pooled && g.nd == 1) {
        note_kernel("row_backward_pool");
        switch (dtype) {
        case SHIFTND_F64: launch_span_backward<f64_t, 1, false, true>(p, sp, active, gw, st); break;
        case SHIFTND_F32: launch_span_backward<f32_t, 1, false, true>(p, sp, active, gw, st); break;
        case SHIFTND_F16: launch_span_backward<f16_t, 1, false, true>(p, sp, active, gw, st); break;
        default: launch_span_backward<bf16_t, 1, false, true>(p, sp, active, gw, st); break;
        }
        return SHIFTND_OK;
    }
    if (pooled) {
        note_kernel("crop_backward_pool");
        switch (dtype) {
        case SHIFTND_F64: launch_span_backward<f64_t, 2, false, true>(p, sp, active, gw, st); break;
        case SHIFTND_F32: launch_span_backward<f32_t, 2, false, true>(p, sp, active, gw, st); break;
        case SHIFTND_F16: launch_span_backward<f16_t, 2, false, true>(p, sp, active, gw, st); break;
        default: launch_span_backward<bf16_t, 2, false, true>(p, sp, active, gw, st); break;
        }
        return SHIFTND_OK;
    }
    if (g.nd == 2 && (g.S[2] * es) % 16 != 0) {   // ragged x rows (4- / 8-byte elements: span_geometry_ok)
        note_kernel("crop_backward_ragged");
        switch (dtype) {
        case SHIFTND_F64: launch_span_backward<f64_t, 2, true>(p, sp, active, gw, st); break;
        case SHIFTND_F32: launch_span_backward<f32_t, 2, true>(p, sp, active, gw, st); break;
        case SHIFTND_F16: launch_span_backward<f16_t, 2, true>(p, sp, active, gw, st); break;
        default: launch_span_backward<bf16_t, 2, true>(p, sp, active, gw, st); break;
        }
        return SHIFTND_OK;
    }
    note_kernel(g.nd == 2 ? "crop_backward" : "row_backward");
#define SHIFTND_SPAN_T(TT) (g.nd == 1 ? launch_span_backward<TT, 1>(p, sp, active, gw, st) : launch_span_backward<TT, 2>(p, sp, active, gw, st))
    switch (dtype) {
    case SHIFTND_F32: SHIFTND_SPAN_T(f32_t); break;
    case SHIFTND_F64: SHIFTND_SPAN_T(f64_t); break;
    case SHIFTND_F16: SHIFTND_SPAN_T(f16_t); break;
    default: SHIFTND_SPAN_T(bf16_t); break;
    }
#undef SHIFTND_SPAN_T
    return SHIFTND_OK;
}

}  // namespace shiftnd
