// shiftnd_bytes.hip -- gather forward of 1-byte elements on small planes (the quantized int8 / uint8 forward of
// BASELINE config 4: N128 C512 56x56), gfx950 (MI355X).
//
// A 56-byte row is not a whole number of 16-byte pieces, so the row-chunk kernels fall back to 8-byte chunks at byte
// alignment and spend ~46 VALU instructions per 8 bytes (plane_gather_forward<1, 8, 2>: 0.124 ms = 41 % of the HBM
// peak, profiles/r02_c4_before_*).  But a PLANE (3136 bytes) is a whole number of pieces, 16-byte aligned, and the
// shift is one number per channel: with D = shift_row * row_bytes + shift_col the output byte at plane offset o is the
// source byte at offset o - D wherever it is not fill -- across row boundaries too.  So
//   * one workgroup = one channel x `ppw` batch entries; it loads its source planes with aligned 16-byte loads
//     (every byte once, all loads issued up front), parks them in LDS, and
//   * builds, while the loads are in flight, a per-channel table with one entry per 16-byte OUTPUT piece of a plane:
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

struct BytesParams {
    const uint8_t *x;
    uint8_t *out;
    const void *w;
    int64_t wzp;
    uint32_t fill4;      // the fill byte in all four byte lanes
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int plane_bytes, npc;   // bytes / 16-byte pieces per plane
    int ppw, groups;        // planes per workgroup, workgroups per channel
    int pitch;              // LDS bytes per plane (plane + guard)
    int use_table;          // non-zero padding: the byte-by-byte source table is built
    unsigned xcd_blocks;
    FastDiv d_npc, d_S2, d_S12, d_C;
    FastDiv d_per[3];    // divide by the padding period of each dim
};

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
    const int n0 = grp * p.ppw, nn = min(p.ppw, p.N - n0);
    const int total = nn * p.npc;  // pieces of this workgroup

    // ---- all source planes: aligned 16-byte loads, issued before anything else -------------------------------
    // piece P = i * 256 + tid of the workgroup = piece k of its plane pl; 32-bit byte offsets from the workgroup's
    // first plane (the host checks ppw * C * plane_bytes < 2^31)
    const int64_t base = (static_cast<int64_t>(n0) * p.C + c) * p.plane_bytes;
    const uint8_t *xb = p.x + base;
    uint8_t *ob = p.out + base;
    const int nstride = p.C * p.plane_bytes;
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
    u4 v[kMaxLoads];
    {
        int pl = pl0, k = k0;
#pragma unroll
        for (int i = 0; i < kMaxLoads; ++i) {
            // (unconditional load: threads past the end re-read piece 0 and drop it)
            const int off = (i * kThreads + static_cast<int>(threadIdx.x)) < total ? pl * nstride + k * 16 : 0;
            // (a plain load: as nontemporal loads these were 10 % slower, 0.092 vs 0.083 ms on C4)
            v[i] = *reinterpret_cast<const u4 *>(xb + off);
            next_piece(pl, k);
        }
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

    // ---- park the planes in LDS --------------------------------------------------------------------------------
    {
        int pl = pl0, k = k0;
#pragma unroll
        for (int i = 0; i < kMaxLoads; ++i) {
            if (i * kThreads + static_cast<int>(threadIdx.x) < total)
                *reinterpret_cast<u4 *>(__builtin_assume_aligned(planes + pl * p.pitch + k * 16, 16)) = v[i];
            next_piece(pl, k);
        }
    }
    __syncthreads();

    // ---- assemble and store the output pieces -----------------------------------------------------------------
    int pl = pl0, k = k0;
#pragma unroll 2
    for (int P = threadIdx.x; P < total; P += kThreads) {
        const int off = poff[k];
        const char *plane = planes + pl * p.pitch;
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
            for (int i = 0; i < 5; ++i) d[i] = dwp[i];
            const uint32_t sb = static_cast<uint32_t>(s0 & 3);
            const u4 mk = pmask[k];
            const uint32_t m[4] = {mk.x, mk.y, mk.z, mk.w};
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t t = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sb);
                w[i] = (t & m[i]) | (p.fill4 & ~m[i]);  // v_bfi_b32
            }
            res = u4{w[0], w[1], w[2], w[3]};
        }
        __builtin_nontemporal_store(res, reinterpret_cast<u4 *>(ob + (pl * nstride + k * 16)));
        next_piece(pl, k);
    }
}

struct BytesPlan {
    int npc, ppw, groups, pitch, use_table;
    size_t lds;
    unsigned grid;
    bool ok;
};

thread_local int g_bytes_tune[3] = {1, 0, 0};  // [0] enabled, [1] planes per workgroup (0 = automatic), [2] LDS bytes for planes

BytesPlan bytes_plan(const Geometry &g) {
    BytesPlan pl{};
    pl.ok = false;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t plane = g.S[0] * g.S[1] * g.S[2];
    if (plane % 16 != 0 || plane > 12288 || plane < 16) return pl;  // (the byte table holds int16 offsets)
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
    if (ppw * g.C * plane >= (1LL << 31)) return pl;  // 32-bit byte offsets inside a workgroup
    pl.ppw = static_cast<int>(ppw);
    pl.groups = static_cast<int>((g.N + ppw - 1) / ppw);
    pl.lds = fixed + static_cast<size_t>(pl.ppw) * pl.pitch;
    if (pl.lds > 64 * 1024) return pl;
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
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
    if (knob >= 0 && knob < 3) g_bytes_tune[knob] = value;
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
    p.pitch = pl.pitch;
    p.use_table = pl.use_table;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_npc = make_fastdiv(static_cast<uint32_t>(pl.npc));
    p.d_S2 = make_fastdiv(static_cast<uint32_t>(g.S[2]));
    p.d_S12 = make_fastdiv(static_cast<uint32_t>(g.S[1] * g.S[2]));
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    for (int d = 0; d < 3; ++d) p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    note_kernel("bytes_gather_forward");
    hipLaunchKernelGGL(bytes_gather_forward, dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
