// shiftnd_cl_tiled.hip -- LDS-tiled sparse-shift / quantized forward for channels-last (NHWC) inputs of 4-byte
// elements, gfx950 (MI355X).  SURVEY section 8f N3.
//
// In NHWC the channels of a pixel are contiguous and every channel has its own shift, so the channel-fastest gather
// of shiftnd_cl.hip sends the 64 lanes of a wave to up to (2 max|shift| + 1)^2 different pixels: one cache-line
// lookup per lane (0.97 TB/s on N16 C256 224x224 fp32).  But every source element is used by exactly one output
// element, so the data can be moved once, through LDS:
//   * a workgroup owns 32 channels (one 128-byte line per pixel) x TW = 32 output columns and walks down the rows of
//     a band; LDS holds a ring of 2 R + 1 source rows of TW + 2 R pixels (R = 3: shifts up to +-3 in each dim take
//     the tiled path per channel, larger ones gather from memory);
//   * per step ONE new source row is staged (whole lines, global -> registers one step ahead -> LDS; 1.19x
//     horizontal halo, no vertical halo) and ONE output row is produced;
//   * the pixel pitch in LDS is 33 words: lanes that read consecutive channels of arbitrary pixels, or consecutive
//     pixels of one channel, hit 32 different banks -- the gather runs at LDS rate;
//   * OUT_CL: the output is channels-last too (the quantized op keeps the format, shifts_quantized.cpp:119-121; a
//     thread keeps one channel and walks the pixels, stores are whole 128-byte pixel lines); otherwise the output is
//     NCHW-contiguous like the reference's float forward (cpu/shifts_cpu.cpp:221): lanes run along the row of one
//     channel and store 128-byte row segments -- the layout change costs nothing extra.
// Periodic padding wraps to the far side of the plane (not in the ring) and is left to shiftnd_cl.hip.
//
// Reference behaviour restated: kernels/shifts_kernels.h:330-400 (nhwdc forward), :574-624 (quantized).
// Roofline: HBM, 2 x 4 bytes per element.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kR = 3;                 // ring half depth = largest |shift| served from LDS
constexpr int kTW = 32;               // output columns per workgroup
constexpr int kCB = 32;               // channels per workgroup (128 bytes)
constexpr int kPW = kTW + 2 * kR;     // staged pixels per row
constexpr int kRing = 2 * kR + 1;     // staged rows
constexpr int kPitch = kCB + 1;       // words per staged pixel
constexpr int kPieces = kPW * (kCB / 4);   // 16-byte pieces per staged row (304)
constexpr int kNP = (kPieces + kThreads - 1) / kThreads;

struct ClTiledParams {
    const uint32_t *x;
    uint32_t *out;
    const void *w;
    int64_t wzp;
    uint32_t fill;
    int wkind, N, C, H, W, pad;
    int out_cl;          // output layout: channels-last (1) or NCHW-contiguous (0)
    int wtiles, cblocks, bands, band_rows;
    FastDiv d_wtiles, d_cblocks, d_bands;
    FastDiv d_perH, d_perW;
};

template <bool OUT_CL>
__global__ __launch_bounds__(kThreads) void cl_tiled_forward(const ClTiledParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t ring[kRing * kPW * kPitch];

    // ---- which tile ---------------------------------------------------------------------------------------------
    unsigned b = blockIdx.x;
    const int wt = static_cast<int>(b - fdiv(b, p.d_wtiles) * p.wtiles);
    b = fdiv(b, p.d_wtiles);
    const int cb = static_cast<int>(b - fdiv(b, p.d_cblocks) * p.cblocks);
    b = fdiv(b, p.d_cblocks);
    const int band = static_cast<int>(b - fdiv(b, p.d_bands) * p.bands);
    const int n = static_cast<int>(fdiv(b, p.d_bands));
    const int w0 = wt * kTW, c0 = cb * kCB;
    const int h0 = band * p.band_rows, h1 = min(p.H, h0 + p.band_rows);
    const int H = p.H, W = p.W, C = p.C;
    const uint32_t *xn = p.x + static_cast<int64_t>(n) * H * W * C;

    // ---- thread -> outputs ----------------------------------------------------------------------------------------
    // OUT_CL: thread = (pixel lane 0..7, channel 0..31): pixels pl + 8 i;  else thread = (column 0..31, channel lane
    // 0..7): channels cl + 8 i.  Either way 4 outputs per step, and per (thread, i) a fixed channel.
    const int lane_a = static_cast<int>(threadIdx.x) & 31, lane_b = static_cast<int>(threadIdx.x) >> 5;
    int ch[4], col[4];      // channel (within the block) and output column (within the tile) of output i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ch[i] = OUT_CL ? lane_a : lane_b + 8 * i;
        col[i] = OUT_CL ? lane_b + 8 * i : lane_a;
    }
    // per output: canonical shifts, the source column (constant over the rows), path
    int csh[4], xoff[4];    // row shift; LDS word offset of the source pixel within a staged row, or -1 (fill), or -2 (far)
    int gcol[4];            // far path: source column in the image (or -1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ch[i];
        csh[i] = 0;
        xoff[i] = -1;
        gcol[i] = -1;
        if (c < C && w0 + col[i] < W) {
            const int sh = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * 2 + 0), H, p.pad, p.d_perH);
            const int sw = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * 2 + 1), W, p.pad, p.d_perW);
            csh[i] = sh;
            const int sx = W == 1 ? 0 : fold_index(w0 + col[i] - sw, W, p.pad);  // size-1 dims ignore the shift
            gcol[i] = sx;
            // canon_shift returns the non-negative representative for the reflecting paddings: look at the signed one
            const int perH = map_period(H, p.pad), perW = map_period(W, p.pad);
            const int sh_s = (perH && 2 * sh > perH) ? sh - perH : sh, sw_s = (perW && 2 * sw > perW) ? sw - perW : sw;
            const bool near = sh_s >= -kR && sh_s <= kR && sw_s >= -kR && sw_s <= kR;
            if (sx >= 0) xoff[i] = near ? (sx - (w0 - kR)) * kPitch + ch[i] : -2;
        }
    }

    // ---- staging: 16-byte pieces of source row y: pixel w0 - R + px, channels c0 + 4 q .. ---------------------------
    int ppx[kNP], pq[kNP];
#pragma unroll
    for (int k = 0; k < kNP; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        ppx[k] = q < kPieces ? q >> 3 : -1;
        pq[k] = q & 7;
        const int gx = w0 - kR + ppx[k];
        if (ppx[k] >= 0 && (gx < 0 || gx >= W || c0 + pq[k] * 4 >= C)) ppx[k] = -1;  // outside the image: never read
    }
    constexpr int kDepth = 3;  // rows of staging in flight (a workgroup moves only ~5 KB per row)
    u4 pvs[kDepth][kNP];
    auto load_row = [&](int y, u4 (&pv)[kNP]) {  // unconditional loads (clamped addresses): see shiftnd_slide.hip
        const int yy = y < 0 ? 0 : (y >= H ? H - 1 : y);
#pragma unroll
        for (int k = 0; k < kNP; ++k) {
            const int gx = ppx[k] >= 0 ? w0 - kR + ppx[k] : w0 < W ? w0 : 0;
            const int cc = ppx[k] >= 0 ? c0 + pq[k] * 4 : 0;
            const uint32_t *src = xn + (static_cast<int64_t>(yy) * W + gx) * C + cc;
            // (C need not be a multiple of 4 in general; the host only routes C % 4 == 0 here)
            pv[k] = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(src, 16));
        }
    };
    auto store_row = [&](int y, const u4 (&pv)[kNP]) {
        const int slot = (y % kRing + kRing) % kRing;
#pragma unroll
        for (int k = 0; k < kNP; ++k) {
            if (ppx[k] >= 0) {
                uint32_t *d = ring + (slot * kPW + ppx[k]) * kPitch + pq[k] * 4;
                d[0] = pv[k].x;
                d[1] = pv[k].y;
                d[2] = pv[k].z;
                d[3] = pv[k].w;
            }
        }
    };

    // rows h0 - R .. h0 + R - 1 first (all loads, then all stores), then one row per step
    {
        u4 pre[2 * kR][kNP];
#pragma unroll
        for (int r = 0; r < 2 * kR; ++r) load_row(h0 - kR + r, pre[r]);
#pragma unroll
        for (int r = 0; r < 2 * kR; ++r) {
            const int y = h0 - kR + r;
            if (y >= 0 && y < H) store_row(y, pre[r]);
        }
    }
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_row(h0 + kR + d, pvs[d]);
    // per output: running ring slot of the interior source row h - s (s = signed shift), running output pointer
    int srow[4], slot[4];
    uint32_t *optr[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int perH = map_period(H, p.pad);
        srow[i] = (perH && 2 * csh[i] > perH) ? csh[i] - perH : csh[i];
        slot[i] = ((h0 - srow[i]) % kRing + kRing) % kRing;
        live[i] = w0 + col[i] < W && c0 + ch[i] < C;
        optr[i] = OUT_CL ? p.out + ((static_cast<int64_t>(n) * H + h0) * W + w0 + col[i]) * C + c0 + ch[i]
                         : p.out + ((static_cast<int64_t>(n) * C + c0 + ch[i]) * H + h0) * W + w0 + col[i];
    }
    const int ostep = OUT_CL ? W * C : W;
    auto step = [&](int h, u4 (&pv)[kNP]) {
        __syncthreads();  // everybody is done with the slot that row h + R replaces (row h - R - 1)
        if (h + kR < H) store_row(h + kR, pv);
        __syncthreads();
        if (h + kDepth < h1) load_row(h + kDepth + kR, pv);  // in flight while this and the next rows are produced
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t v = p.fill;
            int sy = h - srow[i], sl = slot[i];
            if (static_cast<unsigned>(sy) >= static_cast<unsigned>(H)) {  // outside the image: through the padding map
                sy = H == 1 ? 0 : fold_index(h - csh[i], H, p.pad);
                sl = sy % kRing;
            }
            if (H == 1) {
                sy = 0;
                sl = 0;
            }
            if (sy >= 0 && xoff[i] >= 0) v = ring[sl * (kPW * kPitch) + xoff[i]];
            else if (sy >= 0 && xoff[i] == -2) v = xn[(static_cast<int64_t>(sy) * W + gcol[i]) * C + c0 + ch[i]];
            if (live[i]) *optr[i] = v;
            optr[i] += ostep;
            slot[i] = slot[i] + 1 == kRing ? 0 : slot[i] + 1;
        }
    };
    for (int hb = h0; hb < h1; hb += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d)
            if (hb + d < h1) step(hb + d, pvs[d]);
    }
}

thread_local int g_cl_tiled_tune[2] = {1, 0};  // [0] enabled, [1] rows per band (0 = automatic)

bool dense_channels_last_2d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    // normalised strides N, C, d0, d1, inner with d0 of size 1
    return st[1] == 1 && st[4] == g.C && (sz[1] == 1 || st[3] == g.C * sz[2]) && (g.N == 1 || st[0] == g.C * sz[1] * sz[2]);
}
bool contiguous_2d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[4] == 1 && (sz[1] == 1 || st[3] == sz[2]) && st[1] == sz[1] * sz[2] && (g.N == 1 || st[0] == g.C * sz[1] * sz[2]);
}

}  // namespace

void cl_tiled_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 2) g_cl_tiled_tune[knob] = value;
}

// 2-D, 4-byte elements, pure gather (sparse shift / quantized), no crop, not periodic, dense channels-last input with
// C a multiple of 4, 16-byte aligned; output dense channels-last or NCHW-contiguous
bool cl_tiled_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    if (!g_cl_tiled_tune[0] || g.nd != 2 || dtype_size(dtype) != 4 || g.pad == 2) return false;
    if (g.active && dtype <= SHIFTND_BF16) return false;
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return false;
    if (g.C < 4 || g.C % 4 != 0 || g.S[1] >= (1 << 20) || g.S[2] >= (1 << 20) || g.N >= (1LL << 24) || g.C >= (1 << 24)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 4 != 0) return false;
    if (!dense_channels_last_2d(g.xs, g, g.S)) return false;
    return dense_channels_last_2d(g.os, g, g.O) || contiguous_2d(g.os, g, g.O);
}

int cl_tiled_forward(const Geometry &g, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                     hipStream_t st) {
    ClTiledParams p{};
    p.x = static_cast<const uint32_t *>(x);
    p.out = static_cast<uint32_t *>(out);
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = static_cast<uint32_t>(fill_bits);
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.H = static_cast<int>(g.S[1]);
    p.W = static_cast<int>(g.S[2]);
    p.pad = g.pad;
    p.out_cl = dense_channels_last_2d(g.os, g, g.O) ? 1 : 0;
    p.wtiles = (p.W + kTW - 1) / kTW;
    p.cblocks = (p.C + kCB - 1) / kCB;
    // bands along H: enough workgroups (>= ~4096), at least 8 R rows per band (the ring warm-up is 2 R rows)
    const int64_t base = static_cast<int64_t>(p.N) * p.wtiles * p.cblocks;
    int64_t bands = g_cl_tiled_tune[1] > 0 ? (p.H + g_cl_tiled_tune[1] - 1) / g_cl_tiled_tune[1] : (4096 + base - 1) / base;
    const int64_t max_bands = p.H / (8 * kR) > 0 ? p.H / (8 * kR) : 1;
    if (g_cl_tiled_tune[1] <= 0 && bands > max_bands) bands = max_bands;
    if (bands < 1) bands = 1;
    p.band_rows = static_cast<int>((p.H + bands - 1) / bands);
    p.bands = (p.H + p.band_rows - 1) / p.band_rows;
    const int64_t grid = base * p.bands;
    if (grid >= (1LL << 31)) return SHIFTND_ERR_TOO_LARGE;
    p.d_wtiles = make_fastdiv(static_cast<uint32_t>(p.wtiles));
    p.d_cblocks = make_fastdiv(static_cast<uint32_t>(p.cblocks));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_perH = make_fastdiv(static_cast<uint32_t>(map_period(p.H, p.pad)));
    p.d_perW = make_fastdiv(static_cast<uint32_t>(map_period(p.W, p.pad)));
    note_kernel("cl_tiled_forward");
    if (p.out_cl) hipLaunchKernelGGL((cl_tiled_forward<true>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((cl_tiled_forward<false>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
