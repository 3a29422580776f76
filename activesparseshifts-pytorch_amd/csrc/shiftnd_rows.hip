// shiftnd_rows.hip -- sparse-shift / quantized forward of contiguous tensors whose rows are whole 16-byte pieces, for
// 1- and 2-byte elements, gfx950 (MI355X): rows through LDS with register prefetch.
//
// An output row is ONE source row (map1[b]) displaced by the channel's inner shift.  With 1- or 2-byte elements the
// displacement is not a whole number of dwords, so the chunk kernels load at byte alignment (sweep_gather_forward on a
// 224 x 224 uint8 tensor: 1.8 TB/s) or stage through LDS-DMA and wait for it (plane_gather_forward_lds: 5.3-6.0 TB/s
// on C5, box-dependent).  Here, as in shiftnd_cl_tiled.hip:
//   * a workgroup owns one channel and a run of (batch entry [, depth], band of rows) units; a step = R rows (R x 16-byte
//     pieces per row = 256 threads); every thread loads ITS aligned piece of its source row kDepth steps ahead into
//     registers, parks it in one of two LDS row tiles (one barrier per step), and assembles its aligned OUTPUT piece from
//     five dwords at the displaced position (v_alignbyte_b32 for the sub-dword part);
//   * every memory instruction of the step loop is unconditional (raw-buffer offsets out of range for rows that are
//     fill / past the band), so the wait counts are exact and kDepth steps of loads stay in flight;
//   * the two pieces at the ends of a row, where columns are fill or (non-zero paddings) wrap / reflect, go element by
//     element through the column map.
// Reference behaviour restated: kernels/shifts_kernels.h:156-220 (forward, sparse shift), :532-571 (quantized).
// Roofline: HBM, 2 s bytes per element.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kRowGuard = 32;   // bytes in front of / behind every staged row (windows reach 16 before / 20 behind)
#ifndef ROWS_DEPTH
#define ROWS_DEPTH 3
#endif

struct RowsParams {
    const char *x;
    char *out;
    const void *w;
    int64_t wzp;
    uint32_t fill4;      // the fill element replicated over a dword
    int wkind, N, C, nd, pad;
    int S[3], wcol[3];
    int cpr, R;          // 16-byte pieces per row, rows per step
    int BR, bands;       // rows per band (a multiple of R except the last band), bands per plane
    int units, upw;      // N * S0 * bands, units per workgroup
    int map_entries;
    unsigned xcd_blocks;
    FastDiv d_cpr, d_bands, d_S0, d_C;
    FastDiv d_per[3];
};

template <int ES>
__global__ __launch_bounds__(kThreads) void rows_gather_forward(const RowsParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    using EL = typename raw_t<ES>::type;
    constexpr int E = 16 / ES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S0 = p.S[0], S1 = p.S[1], S2 = p.S[2], R = p.R;
    const int RB = S2 * ES, pitch = RB + kRowGuard;
    int *maps = reinterpret_cast<int *>(smem);
    const int *m0 = maps, *m1 = m0 + S0 + 1, *m2 = m1 + S1 + 1;
    char *tiles = smem + ((static_cast<size_t>(p.map_entries) * sizeof(int) + 15) & ~static_cast<size_t>(15));
    const int tile_bytes = R * pitch + kRowGuard;

    const unsigned bid = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int grp = fdiv(bid, p.d_C), c = static_cast<int>(bid) - grp * p.C;
    const int u0 = grp * p.upw, nu = min(p.upw, p.units - u0);

    int64_t sh[3];
    gather_shifts3(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd, p.wcol, sh);
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d] = p.wcol[d] >= 0 ? sh[d] : 0;
    build_maps(maps, p.S, sh, -1, p.pad, p.d_per);
    {   // the guards are never written again: zero both tiles once
        u4 *z = reinterpret_cast<u4 *>(tiles);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = threadIdx.x; i < 2 * tile_bytes / 16; i += kThreads) z[i] = zero;
    }
    __syncthreads();

    // ---- this thread: row slot tr of a step, 16-byte piece tc of the row (source AND output) -----------------------------
    const int tr = fdiv(threadIdx.x, p.d_cpr), tc = static_cast<int>(threadIdx.x) - tr * p.cpr;
    const bool worker = tr < R;
    const int j0 = tc * E;
    // the output piece's columns: affine (every VALID column is j - shift; the others -- beyond the row with zeros padding
    // -- are dropped by a byte mask and replaced by the fill value) or, where a padding wraps or reflects inside the piece,
    // element by element.  With zeros padding no piece takes the element path.
    int cm[E];
    int base = 0;
    bool found = false;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        cm[e] = worker ? m2[j0 + e] : -1;
        if (!found && cm[e] >= 0) {
            base = cm[e] - e;
            found = true;
        }
    }
    bool affine = worker;
    uint32_t mk[4] = {0u, 0u, 0u, 0u};   // bytes of the piece that come from the source row
#pragma unroll
    for (int e = 0; e < E; ++e) {
        affine = affine && (cm[e] < 0 || cm[e] == base + e);
        if (cm[e] >= 0) mk[(e * ES) >> 2] |= (ES == 1 ? 0xffu : 0xffffu) << (((e * ES) & 3) * 8);
    }
    const int s0 = affine ? base * ES : 0;                  // byte offset of the source window in the staged row (>= -15)
    // The window starts at the same byte of its aligned 16-byte piece in every affine piece of the workgroup (s0 = 16 tc -
    // shift * ES): two aligned 16-byte LDS reads (no bank conflicts between the lanes of a row; five dword reads at a lane
    // stride of 16 bytes conflict four ways) and a uniform choice of the five dwords that hold the window.
    const int wq = kRowGuard + (s0 & ~15);
    int phase = 0;
    {
        const unsigned long long am = __ballot(affine && found);
        if (am) phase = __builtin_amdgcn_readlane(s0 & 15, static_cast<int>(__builtin_ctzll(am)));
    }
    const uint32_t wsh = static_cast<uint32_t>(phase & 3);
    const int wk = phase >> 2;

    const int64_t plane_elems = static_cast<int64_t>(S1) * S2;
    const int64_t chan_bytes = static_cast<int64_t>(S0) * plane_elems * ES;   // one (n, c) volume
    // buffer resources over the channel's volumes of one batch entry are rebuilt per unit (uniform); offsets < 2^31 (host)
    constexpr int kDepth = ROWS_DEPTH;
    u4 pv[kDepth];
    // step s of the workgroup = (unit k, row group gi of its band): two cursors walk the steps -- one for the step being
    // produced, one kDepth steps ahead for the loads -- and keep everything that changes only with the unit (batch entry,
    // depth, band: uniform values) out of the per-step arithmetic
    const int gpb = (p.BR + R - 1) / R;                      // row groups per full band
    const int nsteps = nu * gpb;
    struct Cursor {
        int k, gi;             // unit of the workgroup, row group of the band
        int b0, bend;          // first row of the group, end of the band's rows
        int64_t vol;           // byte offset of the (n, c) volume
        uint32_t sp, dp;       // byte offsets of the source / output plane in the volume; sp out of range: fill plane
    };
    auto enter_unit = [&](Cursor &cu) {   // (uniform)
        const int u = u0 + (cu.k < nu ? cu.k : 0);
        const int t1 = fdiv(u, p.d_bands), band = u - t1 * p.bands;     // t1 = n * S0 + a
        const int n = fdiv(t1, p.d_S0), a = t1 - n * S0;
        const int ra = m0[a];
        cu.b0 = band * p.BR;
        cu.bend = cu.k < nu ? min(S1, (band + 1) * p.BR) : 0;
        cu.vol = (static_cast<int64_t>(n) * p.C + c) * chan_bytes;
        cu.sp = ra >= 0 ? static_cast<uint32_t>(static_cast<int64_t>(ra) * S1 * RB) : 0x80000000u;
        cu.dp = static_cast<uint32_t>(static_cast<int64_t>(a) * S1 * RB);
    };
    auto advance = [&](Cursor &cu) {
        cu.b0 += R;
        if (++cu.gi == gpb) {
            cu.gi = 0;
            ++cu.k;
            enter_unit(cu);
        }
    };
    // this thread's piece offsets for the cursor's step (out of range: nothing to load / store), and whether the row is fill
    auto piece = [&](const Cursor &cu, uint32_t &soff, uint32_t &doff, bool &fillrow) {
        const int b = cu.b0 + tr;
        const bool live = worker && b < cu.bend;
        const int rb = live ? m1[b] : -1;
        fillrow = rb < 0 || cu.sp == 0x80000000u;
        soff = (live && !fillrow) ? cu.sp + static_cast<uint32_t>(rb * RB + tc * 16) : 0x80000000u;
        doff = live ? cu.dp + static_cast<uint32_t>(b * RB + tc * 16) : 0x80000000u;
    };
    Cursor cur{0, 0, 0, 0, 0, 0u, 0u}, ahead{0, 0, 0, 0, 0, 0u, 0u};
    enter_unit(cur);
    enter_unit(ahead);
    auto issue = [&](u4 &v) {   // the loads of the `ahead` cursor's step, then on to the next step
        uint32_t soff, doff;
        bool fr;
        piece(ahead, soff, doff, fr);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.x) + ahead.vol, 0, 0x7ffffffc, 0x00020000);
        v = __builtin_amdgcn_raw_buffer_load_b128(r, soff, 0, 0);
        advance(ahead);
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d) issue(pv[d]);
    {   // kDepth dropped stores: the loop's entry path then has as many operations behind its loads as the back edge
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
        const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int d = 0; d < kDepth; ++d) __builtin_amdgcn_raw_buffer_store_b128(z, r, 0x80000000u + d * 16, 0, 0);
    }
    auto step = [&](int s, u4 &v) {
        char *tile = tiles + (s & 1) * tile_bytes;
        if (worker) *reinterpret_cast<u4 *>(__builtin_assume_aligned(tile + tr * pitch + kRowGuard + tc * 16, 16)) = v;
        __syncthreads();
        uint32_t soff, doff;
        bool fillrow;
        piece(cur, soff, doff, fillrow);
        char *dst = p.out + cur.vol;
        advance(cur);
        issue(v);
        const char *row = tile + (worker ? tr : 0) * pitch;
        uint32_t w[4];
        if (affine) {
            const u4 q0 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(row + wq, 16));
            const u4 q1 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(row + wq + 16, 16));
            const uint32_t D[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            switch (wk) {   // (uniform)
            case 0:
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbyte(D[i + 1], D[i], wsh);
                break;
            case 1:
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbyte(D[i + 2], D[i + 1], wsh);
                break;
            case 2:
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbyte(D[i + 3], D[i + 2], wsh);
                break;
            default:
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbyte(D[i + 4], D[i + 3], wsh);
                break;
            }
        } else {
            const EL *re = reinterpret_cast<const EL *>(row + kRowGuard);
            const EL fe = static_cast<EL>(p.fill4);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const EL val = cm[e] >= 0 ? re[cm[e] > 0 ? cm[e] : 0] : fe;
                w[(e * ES) >> 2] |= static_cast<uint32_t>(val) << (((e * ES) & 3) * 8);
            }
        }
        if (affine) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (w[i] & mk[i]) | (p.fill4 & ~mk[i]);   // v_bfi_b32
        }
        const u4 res = fillrow ? u4{p.fill4, p.fill4, p.fill4, p.fill4} : u4{w[0], w[1], w[2], w[3]};
        const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7ffffffc, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(res, ores, doff, 0, 2);   // nontemporal
    };
    int s = 0;
    for (; s + kDepth <= nsteps; s += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) step(s + d, pv[d]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)
        if (s + d < nsteps) step(s + d, pv[d]);
}

struct RowsPlan {
    int cpr, R, BR, bands, units, upw, groups, map_entries;
    size_t lds;
    unsigned grid;
    bool ok;
};

thread_local int g_rows_tune[3] = {1, 0, 0};  // [0] which element sizes take the kernel (bit 0: 1-byte, bit 1: 2-byte), [1] rows per band, [2] workgroups wanted

RowsPlan rows_plan(const Geometry &g, int es) {
    RowsPlan pl{};
    pl.ok = false;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return pl;
    const int64_t RB = g.S[2] * es;
    if (RB % 16 != 0 || RB / 16 > kThreads || RB < 16) return pl;
    if (g.N >= (1LL << 30) || g.C >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return pl;
    if (g.S[0] * g.S[1] * RB >= (1LL << 31)) return pl;   // buffer offsets inside one (n, c) volume
    pl.cpr = static_cast<int>(RB / 16);
    pl.R = kThreads / pl.cpr;
    if (pl.R > g.S[1]) pl.R = static_cast<int>(g.S[1]);
    pl.map_entries = static_cast<int>(g.S[0] + g.S[1] + g.S[2] + 3);
    const size_t map_bytes = (static_cast<size_t>(pl.map_entries) * sizeof(int) + 15) & ~static_cast<size_t>(15);
    pl.lds = map_bytes + 2 * (static_cast<size_t>(pl.R) * (RB + kRowGuard) + kRowGuard);
    if (pl.lds > 60 * 1024) return pl;
    // bands: rows per band a multiple of R; enough units for ~4096 workgroups, at least ~8 steps per band
    const int64_t wanted = g_rows_tune[2] > 0 ? g_rows_tune[2] : 4096;
    int64_t br = g_rows_tune[1] > 0 ? g_rows_tune[1] : g.S[1];
    if (g_rows_tune[1] <= 0) {
        const int64_t planes = g.N * g.S[0];
        int64_t bands = (wanted + g.C * planes - 1) / (g.C * planes);
        const int64_t max_bands = g.S[1] / (8 * pl.R) > 0 ? g.S[1] / (8 * pl.R) : 1;
        if (bands > max_bands) bands = max_bands;
        if (bands < 1) bands = 1;
        br = (g.S[1] + bands - 1) / bands;
    }
    br = (br + pl.R - 1) / pl.R * pl.R;
    pl.BR = static_cast<int>(br);
    pl.bands = static_cast<int>((g.S[1] + br - 1) / br);
    const int64_t units = g.N * g.S[0] * pl.bands;
    if (units >= (1LL << 30)) return pl;
    pl.units = static_cast<int>(units);
    int64_t groups = (wanted + g.C - 1) / g.C;
    if (groups > units) groups = units;
    if (groups < 1) groups = 1;
    pl.upw = static_cast<int>((units + groups - 1) / groups);
    pl.groups = static_cast<int>((units + pl.upw - 1) / pl.upw);
    const int64_t grid = static_cast<int64_t>(pl.groups) * g.C;
    if (grid >= (1LL << 31)) return pl;
    pl.grid = static_cast<unsigned>(grid);
    pl.ok = true;
    return pl;
}

bool contiguous5r(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

}  // namespace

void rows_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 3) g_rows_tune[knob] = value;
}

// sparse-shift / quantized forward, 1- or 2-byte elements (knob 28 selects which), contiguous, no crop, rows of whole
// 16-byte pieces, 16-byte aligned tensors
bool rows_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    const int es = dtype_size(dtype);
    if (es > 2 || !((g_rows_tune[0] >> (es - 1)) & 1) || (g.active && dtype <= SHIFTND_BF16)) return false;
    if (!contiguous5r(g.xs, g.N, g.C, g.S) || !contiguous5r(g.os, g.N, g.C, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16) return false;
    return rows_plan(g, es).ok;
}

int rows_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                 hipStream_t st) {
    const int es = dtype_size(dtype);
    const RowsPlan pl = rows_plan(g, es);
    if (!pl.ok) return SHIFTND_ERR_INVALID_ARGUMENT;
    RowsParams p{};
    p.x = static_cast<const char *>(x);
    p.out = static_cast<char *>(out);
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill4 = es == 1 ? static_cast<uint32_t>(fill_bits & 0xff) * 0x01010101u : static_cast<uint32_t>(fill_bits & 0xffff) * 0x00010001u;
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.cpr = pl.cpr;
    p.R = pl.R;
    p.BR = pl.BR;
    p.bands = pl.bands;
    p.units = pl.units;
    p.upw = pl.upw;
    p.map_entries = pl.map_entries;
    p.xcd_blocks = pl.grid % 8 == 0 ? pl.grid / 8 : 0;
    p.d_cpr = make_fastdiv(static_cast<uint32_t>(pl.cpr));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(pl.bands));
    p.d_S0 = make_fastdiv(static_cast<uint32_t>(p.S[0]));
    p.d_C = make_fastdiv(static_cast<uint32_t>(p.C));
    note_kernel("rows_gather_forward");
    if (es == 1) hipLaunchKernelGGL(rows_gather_forward<1>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    else hipLaunchKernelGGL(rows_gather_forward<2>, dim3(pl.grid), dim3(kThreads), pl.lds, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
