// shiftnd_bytes.hip -- gather forward of 1-byte elements on small planes (the quantized int8 / uint8 forward of
// BASELINE config 4: N128 C512 56x56), gfx950 (MI355X).
//
// A 56-byte row is not a whole number of 16-byte pieces, so the row-chunk kernels fall back to 8-byte chunks at byte
// alignment and spend ~46 VALU instructions per 8 bytes (plane_gather_forward<1, 8, 2>: 0.124 ms = 41 % of the HBM
// peak, profiles/r02_c4_before_*).  But a PLANE (3136 bytes) is a whole number of pieces, 16-byte aligned, and the
// shift is one number per channel: with D = shift_row * row_bytes + shift_col the output byte at plane offset o is the
// source byte at offset o - D wherever it is not fill -- across row boundaries too.  So
//   * one workgroup = one channel x `ppw * rpw` batch entries, in `rpw` rounds of `ppw` planes; per round it loads its
//     source planes with aligned 16-byte loads (every byte once; the NEXT round's loads are issued before this round
//     is assembled, and every store is unconditional -- an out-of-range buffer offset where there is nothing to
//     store -- so that the compiler waits for the loads only, never for the stores), parks them in LDS, and
//   * builds ONCE, while the first loads are in flight, a per-channel table with one entry per 16-byte OUTPUT piece of a plane:
//     the (constant) source offset of its valid bytes and a byte mask of the valid ones; then
//   * every thread assembles aligned 16-byte output pieces: five dwords from LDS at the source offset, one funnel
//     shift per dword (v_alignbyte_b32), one v_bfi_b32 per dword against the mask to drop in the fill value (the
//     input's zero point, kernels/shifts_kernels.h:569), one 16-byte store.
// Pieces whose valid bytes do not come from one contiguous source run (the wrapped / reflected edges of the
// non-zero paddings) take a byte-by-byte path through a per-channel source-offset table.
//
// Reference behaviour restated: kernels/shifts_kernels.h:532-571 (shift = int_repr(w) - w.zero_point, :553-555),
// quantized/shifts_quantized.cpp:107-130.  Roofline: HBM, 2 bytes per element.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kMaxLoads = 10;      // 16-byte pieces a thread loads (ppw * pieces per plane <= kMaxLoads * 256)
constexpr int kPlaneGuard = 32;    // bytes between planes in LDS (windows of edge pieces reach 16 before / 20 behind)
constexpr int kAllFill = INT32_MIN, kGeneral = INT32_MIN + 1;
constexpr int kBytesWgs = 1024;    // workgroups wanted when a workgroup runs several rounds

struct BytesParams {
    const uint8_t *x;
    uint8_t *out;
    const void *w;
    int64_t wzp;
    uint32_t fill4;      // the fill byte in all four byte lanes
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int plane_bytes, npc;   // bytes / 16-byte pieces per plane
    int ppw, groups;        // planes per round of a workgroup, workgroups per channel
    int rpw;                // rounds per workgroup: it owns ppw * rpw consecutive planes of its channel
    int pitch;              // LDS bytes per plane (plane + guard)
    int use_table;          // non-zero padding: the byte-by-byte source table is built
    unsigned xcd_blocks;
    FastDiv d_npc, d_S2, d_S12, d_C;
    FastDiv d_per[3];    // divide by the padding period of each dim
};

// NL = 16-byte pieces a thread loads / stores per round (ppw * pieces per plane <= NL * 256).
template <int NL>
__global__ __launch_bounds__(kThreads) void bytes_gather_forward(const BytesParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2];
    // LDS: [guard][planes ...] | masks (npc x 16 B) | offsets (npc ints) | maps | byte table (int16 per plane byte)
    char *planes = smem + kPlaneGuard;
    char *after = smem + kPlaneGuard + p.ppw * p.pitch;
    u4 *pmask = reinterpret_cast<u4 *>(after);
    int *poff = reinterpret_cast<int *>(after + p.npc * 16);
    int *maps = poff + p.npc;
    const int *m0 = maps, *m1 = m0 + S0 + 1, *m2 = m1 + S1 + 1;
    int16_t *tab = reinterpret_cast<int16_t *>(maps + S0 + S1 + S2 + 3);

    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int grp = fdiv(bid, p.d_C), c = static_cast<int>(bid) - grp * p.C;
    const int n0 = grp * p.ppw * p.rpw, nw = min(p.ppw * p.rpw, p.N - n0);   // this workgroup's planes: rounds of ppw

    // ---- source planes of a round: aligned 16-byte loads, issued before anything else ----------------------------
    // piece P = i * 256 + tid of the round = piece k of its plane pl; 32-bit byte offsets from the workgroup's
    // first plane (the host checks ppw * rpw * C * plane_bytes < 2^31)
    const int64_t base = (static_cast<int64_t>(n0) * p.C + c) * p.plane_bytes;
    const uint8_t *xb = p.x + base;
    const int nstride = p.C * p.plane_bytes;
    // every store of the round loop is unconditional (pieces past the end: an offset the hardware drops), so that hipcc
    // can count the stores between the next round's loads and their use, and waits for the loads only
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(p.out + base, 0, 0x7ffffffc, 0x00020000);
    const int q256 = fdiv(kThreads, p.d_npc), r256 = kThreads - q256 * p.npc;
    const int pl0 = fdiv(threadIdx.x, p.d_npc), k0 = static_cast<int>(threadIdx.x) - pl0 * p.npc;
    auto next_piece = [&](int &pl, int &k) {
        pl += q256;
        k += r256;
        if (k >= p.npc) {
            k -= p.npc;
            ++pl;
        }
    };
    u4 v[NL];
    auto load_round = [&](int r) {   // (rounds past the end: every thread re-reads piece 0 and drops it)
        const int total = max(0, min(p.ppw, nw - r * p.ppw)) * p.npc;
        const int rbase = r * p.ppw * nstride;
        int pl = pl0, k = k0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            // (unconditional load: threads past the end re-read piece 0 and drop it)
            const int off = (i * kThreads + static_cast<int>(threadIdx.x)) < total ? rbase + pl * nstride + k * 16 : 0;
            // (a plain load: as nontemporal loads these were 10 % slower, 0.092 vs 0.083 ms on C4)
            v[i] = *reinterpret_cast<const u4 *>(xb + off);
            next_piece(pl, k);
        }
    };
    load_round(0);
    {   // NL dropped stores: the first round then sees the same number of operations behind its loads as every other
        // round (the wait counts of a loop are the minimum over its entry and its back edge)
        const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NL; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, ores, 0x80000000u, 0, 0);
    }

    // ---- per-channel tables (while the loads are in flight) ---------------------------------------------------
    int64_t sh[3];
    gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd, p.wcol, sh);
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d] = p.wcol[d] >= 0 ? sh[d] : 0;
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    __syncthreads();
    // one table entry per 16-byte output piece: walk its bytes through the maps (divisions once per piece)
    for (int k = threadIdx.x; k < p.npc; k += kThreads) {
        uint32_t mk[4] = {0u, 0u, 0u, 0u};
        int off = kAllFill;
        bool general = false;
        int a = fdiv(k * 16, p.d_S12);
        const int r = k * 16 - a * (S1 * S2);
        int b = fdiv(r, p.d_S2), cc = r - b * S2;
        int ra = m0[a], rb = m1[b];
        int rowoff = (ra * S1 + rb) * S2;
        bool rowok = ra >= 0 && rb >= 0;
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) {
            const int rc = m2[cc];
            const int s = (rowok && rc >= 0) ? rowoff + rc : -1;  // source offset of output byte 16 k + bb, or fill
            if (p.use_table) tab[k * 16 + bb] = static_cast<int16_t>(s);
            if (s >= 0) {
                mk[bb >> 2] |= 0xffu << ((bb & 3) * 8);
                const int d = s - (k * 16 + bb);
                if (off == kAllFill) off = d;
                else if (off != d) general = true;
            }
            if (++cc == S2) {  // next row (entries S1 / S0 of the maps exist: the walk may step just past the plane)
                cc = 0;
                if (++b == S1 && S0 > 1) {
                    b = 0;
                    ra = m0[++a];
                }
                rb = m1[b];
                rowoff = (ra * S1 + rb) * S2;
                rowok = ra >= 0 && rb >= 0;
            }
        }
        pmask[k] = u4{mk[0], mk[1], mk[2], mk[3]};
        poff[k] = general ? kGeneral : off;
    }

    for (int r = 0; r * p.ppw < nw; ++r) {
        const int total = min(p.ppw, nw - r * p.ppw) * p.npc;  // pieces of this round
        // ---- park the planes in LDS ----------------------------------------------------------------------------
        {
            int pl = pl0, k = k0;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (i * kThreads + static_cast<int>(threadIdx.x) < total)
                    *reinterpret_cast<u4 *>(__builtin_assume_aligned(planes + pl * p.pitch + k * 16, 16)) = v[i];
                next_piece(pl, k);
            }
        }
        __syncthreads();
        load_round(r + 1);  // in flight while this round is assembled

        // ---- assemble and store the output pieces ----------------------------------------------------------------
        const uint32_t rbase = static_cast<uint32_t>(r * p.ppw) * static_cast<uint32_t>(nstride);
        int pl = pl0, k = k0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool mine = i * kThreads + static_cast<int>(threadIdx.x) < total;
            const int off = mine ? poff[k] : kAllFill;
            const char *plane = planes + (mine ? pl * p.pitch : 0);
            u4 res = {p.fill4, p.fill4, p.fill4, p.fill4};
            if (off == kGeneral) {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int s = tab[k * 16 + b];
                    const uint32_t byte = s >= 0 ? static_cast<uint8_t>(plane[s]) : (p.fill4 & 0xffu);
                    w[b >> 2] |= byte << ((b & 3) * 8);
                }
                res = u4{w[0], w[1], w[2], w[3]};
            } else if (off != kAllFill) {
                const int s0 = k * 16 + off;  // source byte of the piece's byte 0: >= -15, < plane_bytes
                const uint32_t *dwp = reinterpret_cast<const uint32_t *>(plane + (s0 & ~3));
                uint32_t d[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) d[j] = dwp[j];
                const uint32_t sb = static_cast<uint32_t>(s0 & 3);
                const u4 mk = pmask[k];
                const uint32_t m[4] = {mk.x, mk.y, mk.z, mk.w};
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sb);
                    w[j] = (t & m[j]) | (p.fill4 & ~m[j]);  // v_bfi_b32
                }
                res = u4{w[0], w[1], w[2], w[3]};
            }
            const uint32_t ooff = mine ? rbase + static_cast<uint32_t>(pl * nstride + k * 16) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(res, ores, ooff, 0, 2);  // nontemporal
            next_piece(pl, k);
        }
        __syncthreads();  // the next round parks its planes over these
    }
}

struct BytesPlan {
    int npc, ppw, rpw, nl, groups, pitch, use_table;
    size_t lds;
    unsigned grid;
    bool ok;
};

thread_local int g_bytes_tune[4] = {1, 0, 0, 0};  // [0] enabled, [1] planes per round (0 = automatic), [2] LDS bytes for planes, [3] rounds per workgroup

BytesPlan bytes_plan(const Geometry &g) {
    BytesPlan pl{};
    pl.ok = false;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t plane = g.S[0] * g.S[1] * g.S[2];
    if (plane % 16 != 0 || plane > 16384 || plane < 16) return pl;  // (the byte table holds int16 offsets; 112 x 112 = 12544 still fits)
    if (g.S[0] + g.S[1] + g.S[2] + 3 > 2048) return pl;
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30)) return pl;
    pl.npc = static_cast<int>(plane / 16);
    pl.pitch = static_cast<int>(plane) + kPlaneGuard;
    pl.use_table = g.pad != 0;
    const size_t fixed = kPlaneGuard + static_cast<size_t>(pl.npc) * 20 + static_cast<size_t>(g.S[0] + g.S[1] + g.S[2] + 3) * 4 +
                         (pl.use_table ? static_cast<size_t>(plane) * 2 : 0) + 16;
    // planes per workgroup: as many as kMaxLoads pieces per thread and ~24 KiB of LDS allow, but enough workgroups
    int64_t ppw = static_cast<int64_t>(kMaxLoads) * kThreads / pl.npc;
    const int64_t by_lds = (g_bytes_tune[2] > 0 ? g_bytes_tune[2] : 32 * 1024) / pl.pitch;
    if (ppw > by_lds) ppw = by_lds;
    if (g_bytes_tune[1] > 0) ppw = g_bytes_tune[1];
    if (ppw > g.N) ppw = g.N;
    while (ppw > 1 && g.C * ((g.N + ppw - 1) / ppw) < 4096) --ppw;
    if (g_bytes_tune[1] <= 0 && ppw > 1) {
        // the last workgroup of a channel should not be ragged: within -40 % take the ppw that wastes the fewest plane
        // slots (C4, N = 128: 8 planes per workgroup 0.088 ms, 9 or 10 planes 0.100 ms)
        auto waste = [&](int64_t q) { return static_cast<double>((g.N + q - 1) / q * q) / static_cast<double>(g.N); };
        int64_t best = ppw;
        for (int64_t q = ppw; q >= 1 && q >= ppw - (ppw * 2) / 5; --q)
            if (waste(q) < waste(best) - 1e-9) best = q;
        ppw = best;
    }
    if (ppw < 1 || ppw * pl.npc > static_cast<int64_t>(kMaxLoads) * kThreads) return pl;
    pl.ppw = static_cast<int>(ppw);
    // rounds per workgroup: the per-channel tables are built once per workgroup and the next round's planes load while
    // this round is assembled; as many as leave ~4 workgroups per CU (knob 19).  C4: 1 round 0.080 ms, 2+ rounds 0.067 ms
    // (6.1 TB/s; torch's plain copy of the tensor: 0.077 ms); reflect padding 0.116 -> 0.088 ms with 8 rounds
    int64_t rpw = g_bytes_tune[3] > 0 ? g_bytes_tune[3] : 1;
    if (g_bytes_tune[3] <= 0)
        while (rpw < 64 && g.C * ((g.N + ppw * rpw * 2 - 1) / (ppw * rpw * 2)) >= kBytesWgs) rpw *= 2;
    if (ppw * rpw > g.N) rpw = (g.N + ppw - 1) / ppw;
    if (ppw * rpw * g.C * plane >= (1LL << 31)) return pl;  // 32-bit byte offsets inside a workgroup
    pl.rpw = static_cast<int>(rpw);
    const int64_t need = (ppw * pl.npc + kThreads - 1) / kThreads;
    pl.nl = need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 7 ? 7 : 10));
    pl.groups = static_cast<int>((g.N + ppw * rpw - 1) / (ppw * rpw));
    pl.lds = fixed + static_cast<size_t>(pl.ppw) * pl.pitch;
    if (pl.lds > 64 * 1024) return pl;
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

// =====================================================================================================================
// bytes_block_forward: planes that are NOT whole 16-byte pieces (14 x 14 = 196 bytes, 7 x 7 = 49: the late stages of a
// quantized ResNet-shaped network), which the chunk kernels move element by element (0.7 - 0.9 TB/s).  The planes of
// cb = 16 / gcd(plane bytes, 16) consecutive channels of one batch entry ARE a run of whole, aligned pieces (196 x 4 =
// 49 x 16 = 784 bytes), so a workgroup owns one block of cb channels and a group of batch entries:
//   * once: a table with one int16 per byte of the block -- the byte's source offset inside the block, through the
//     padding map of ITS channel, or -1 for the fill value; every thread keeps the 16 entries of its output piece in
//     registers for the life of the workgroup;
//   * per round of nr batch entries: each thread loads one aligned piece (the next round's before this round is
//     assembled), parks it in LDS, gathers its 16 output bytes through its table entries and stores one aligned piece.
// Any padding, any number of dims; no crop.
// =====================================================================================================================
struct BlockParams {
    const uint8_t *x;
    uint8_t *out;
    const void *w;
    int64_t wzp;
    uint32_t fill;
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int plane_bytes, cb, block_bytes, npb;  // channels per block, bytes / pieces per block
    int nr, npw;                            // batch entries per round, per workgroup
    int cblocks;                            // C / cb
    FastDiv d_npb, d_plane, d_S2, d_S12, d_cblocks;
    FastDiv d_per[3];
};

__global__ __launch_bounds__(kThreads) void bytes_block_forward(const BlockParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [2 x nr block images] | table (int16 per block byte) | canonical shifts (cb x 3 ints)
    char *img = smem;
    const int img_bytes = p.nr * p.block_bytes;
    int16_t *tab = reinterpret_cast<int16_t *>(smem + 2 * img_bytes);
    int *cs = reinterpret_cast<int *>(smem + 2 * img_bytes + ((p.block_bytes * 2 + 15) & ~15));
    const int tid = static_cast<int>(threadIdx.x);
    const int grp = static_cast<int>(fdiv(blockIdx.x, p.d_cblocks)), cblk = static_cast<int>(blockIdx.x) - grp * p.cblocks;
    const int c0 = cblk * p.cb;
    const int n0 = grp * p.npw, nn = min(p.npw, p.N - n0);
    // this thread's piece: batch entry j of a round, piece t of the block
    const int j = static_cast<int>(fdiv(static_cast<uint32_t>(tid), p.d_npb)), t = tid - j * p.npb;
    const bool mine = j < p.nr;
    const int64_t stride_n = static_cast<int64_t>(p.C) * p.plane_bytes;
    const uint8_t *xb = p.x + (static_cast<int64_t>(n0) * p.C + c0) * p.plane_bytes + t * 16;
    uint8_t *ob = p.out + (static_cast<int64_t>(n0) * p.C + c0) * p.plane_bytes + t * 16;
    u4 cur = {0, 0, 0, 0};
    if (mine && j < nn) cur = *reinterpret_cast<const u4 *>(xb + j * stride_n);  // first round, in flight behind the table
    // ---- canonical shifts of the block's channels ---------------------------------------------------------------------
    if (tid < p.cb * 3) {
        const int pl = tid / 3, d = tid - pl * 3;
        int v = 0;
        if (p.wcol[d] >= 0) {
            const int64_t sh = gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c0 + pl) * p.nd + p.wcol[d]);
            v = canon_shift(sh, p.S[d], p.pad, p.d_per[d]);
        }
        cs[tid] = v;
    }
    __syncthreads();
    // ---- the table: source offset of every byte of the block ----------------------------------------------------------
    for (int q = tid; q < p.block_bytes; q += kThreads) {
        const int pl = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_plane));
        const int o = q - pl * p.plane_bytes;
        const int a = static_cast<int>(fdiv(static_cast<uint32_t>(o), p.d_S12));
        const int rem = o - a * p.S[1] * p.S[2];
        const int b = static_cast<int>(fdiv(static_cast<uint32_t>(rem), p.d_S2));
        const int cc = rem - b * p.S[2];
        const int sa = p.S[0] == 1 ? 0 : fold_index(a - cs[pl * 3 + 0], p.S[0], p.pad);
        const int sb = p.S[1] == 1 ? 0 : fold_index(b - cs[pl * 3 + 1], p.S[1], p.pad);
        const int sc = p.S[2] == 1 ? 0 : fold_index(cc - cs[pl * 3 + 2], p.S[2], p.pad);
        tab[q] = (sa < 0 || sb < 0 || sc < 0) ? static_cast<int16_t>(-1)
                                               : static_cast<int16_t>(pl * p.plane_bytes + (sa * p.S[1] + sb) * p.S[2] + sc);
    }
    __syncthreads();
    int16_t src[16];
    {
        const u4 lo = *reinterpret_cast<const u4 *>(tab + (mine ? t : 0) * 16);
        const u4 hi = *reinterpret_cast<const u4 *>(tab + (mine ? t : 0) * 16 + 8);
        __builtin_memcpy(src, &lo, 16);
        __builtin_memcpy(src + 8, &hi, 16);
    }
    // ---- rounds -------------------------------------------------------------------------------------------------------
    int buf = 0;
    for (int r0 = 0; r0 < nn; r0 += p.nr) {
        char *im = img + buf * img_bytes;
        if (mine) *reinterpret_cast<u4 *>(im + j * p.block_bytes + t * 16) = cur;
        const int jn = r0 + p.nr + j;
        u4 nxt = {0, 0, 0, 0};
        if (mine && jn < nn) nxt = *reinterpret_cast<const u4 *>(xb + jn * stride_n);
        __syncthreads();  // (two images: the next round writes the other one, so one barrier per round)
        if (mine && r0 + j < nn) {
            const uint8_t *blk = reinterpret_cast<const uint8_t *>(im + j * p.block_bytes);
            uint32_t o4[4];
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                uint32_t word = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sidx = src[d4 * 4 + k];
                    const uint32_t v = sidx >= 0 ? blk[sidx] : (p.fill & 0xffu);
                    word |= v << (8 * k);
                }
                o4[d4] = word;
            }
            const u4 res = {o4[0], o4[1], o4[2], o4[3]};
            __builtin_nontemporal_store(res, reinterpret_cast<u4 *>(ob + (r0 + j) * stride_n));
        }
        cur = nxt;
        buf ^= 1;
    }
}

struct BlockPlan {
    bool ok = false;
    int cb = 0, block_bytes = 0, npb = 0, nr = 0, npw = 0, groups = 0;
    size_t lds = 0;
    unsigned grid = 0;
};

BlockPlan block_plan(const Geometry &g) {
    BlockPlan pl;
    const int64_t plane = g.S[0] * g.S[1] * g.S[2];
    if (plane < 1 || plane % 16 == 0 || plane > 4096) return pl;  // (whole-piece planes: bytes_gather_forward)
    int gcd = 16;
    while (plane % gcd) gcd /= 2;
    pl.cb = 16 / gcd;
    if (g.C % pl.cb) return pl;
    pl.block_bytes = static_cast<int>(plane * pl.cb);
    if (pl.block_bytes > 16384) return pl;  // int16 table, LDS
    pl.npb = pl.block_bytes / 16;
    if (pl.npb > kThreads) return pl;
    pl.nr = kThreads / pl.npb;
    if (pl.nr > g.N) pl.nr = static_cast<int>(g.N);
    // batch entries per workgroup: whole rounds, ~2048+ workgroups in all when the batch allows
    const int64_t cblocks = g.C / pl.cb;
    int64_t rounds = (g.N + pl.nr - 1) / pl.nr;
    int64_t rpw = rounds;
    while (rpw > 1 && cblocks * ((rounds + rpw - 1) / rpw) < 2048) rpw = (rpw + 1) / 2;
    pl.npw = static_cast<int>(rpw * pl.nr);
    pl.groups = static_cast<int>((g.N + pl.npw - 1) / pl.npw);
    pl.lds = 2 * static_cast<size_t>(pl.nr) * pl.block_bytes + ((static_cast<size_t>(pl.block_bytes) * 2 + 15) & ~static_cast<size_t>(15)) + pl.cb * 3 * sizeof(int);
    if (pl.lds > 64 * 1024) return pl;
    if (g.N * g.C * plane >= (1LL << 40)) return pl;
    const int64_t grid = cblocks * pl.groups;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

bool contiguous5b(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

}  // namespace

void bytes_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 4) g_bytes_tune[knob] = value;
}

bool bytes_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (!g_bytes_tune[0] || dtype_size(dtype) != 1 || g.active) return false;
    if (!contiguous5b(g.xs, g.N, g.C, g.S) || !contiguous5b(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16) return false;
    return bytes_plan(g).ok;
}

int bytes_forward(const Geometry &g, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                  hipStream_t st) {
    const BytesPlan pl = bytes_plan(g);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    BytesParams p{};
    p.x = static_cast<const uint8_t *>(x);
    p.out = static_cast<uint8_t *>(out);
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill4 = static_cast<uint32_t>(fill_bits & 0xff) * 0x01010101u;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
    }
    p.plane_bytes = static_cast<int>(g.S[0] * g.S[1] * g.S[2]);
    p.npc = pl.npc;
    p.ppw = pl.ppw;
    p.groups = pl.groups;
    p.rpw = pl.rpw;
    p.pitch = pl.pitch;
    p.use_table = pl.use_table;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_npc = make_fastdiv(static_cast<uint32_t>(pl.npc));
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(g.S[2]));
    p.d_S12 = make_fastdiv(static_cast<uint32_t>(g.S[1] * g.S[2]));
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    for (int d = 0; d < 3; ++d) p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    note_kernel("bytes_gather_forward");
    switch (pl.nl) {
    case 2: hipLaunchKernelGGL(bytes_gather_forward<2>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    case 4: hipLaunchKernelGGL(bytes_gather_forward<4>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    case 7: hipLaunchKernelGGL(bytes_gather_forward<7>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    default: hipLaunchKernelGGL(bytes_gather_forward<10>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p); break;
    }
    return SHIFTND_OK;
}

// one-byte planes that are not whole 16-byte pieces, no crop, contiguous
bool bytes_block_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (!g_bytes_tune[0] || dtype_size(dtype) != 1 || g.active) return false;
    for (int d = 0; d < 3; ++d)
        if (g.O[d] != g.S[d] || g.L[d] != 0) return false;
    if (!contiguous5b(g.xs, g.N, g.C, g.S) || !contiguous5b(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16) return false;
    return block_plan(g).ok;
}

int bytes_block_forward(const Geometry &g, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                        hipStream_t st) {
    const BlockPlan pl = block_plan(g);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    BlockParams p{};
    p.x = static_cast<const uint8_t *>(x);
    p.out = static_cast<uint8_t *>(out);
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = static_cast<uint32_t>(fill_bits & 0xff);
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.plane_bytes = static_cast<int>(g.S[0] * g.S[1] * g.S[2]);
    p.cb = pl.cb;
    p.block_bytes = pl.block_bytes;
    p.npb = pl.npb;
    p.nr = pl.nr;
    p.npw = pl.npw;
    p.cblocks = static_cast<int>(g.C / pl.cb);
    p.d_npb = make_fastdiv(static_cast<uint32_t>(pl.npb));
    p.d_plane = make_fastdiv(static_cast<uint32_t>(p.plane_bytes));
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(g.S[2]));
    p.d_S12 = make_fastdiv(static_cast<uint32_t>(g.S[1] * g.S[2]));
    p.d_cblocks = make_fastdiv(static_cast<uint32_t>(p.cblocks));
    note_kernel("bytes_block_forward");
    hipLaunchKernelGGL(bytes_block_forward, dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
