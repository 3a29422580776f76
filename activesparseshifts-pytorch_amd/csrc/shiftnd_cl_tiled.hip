// shiftnd_cl_tiled.hip -- LDS-tiled kernels for dense channels-last (NHWC) tensors, 2-D, gfx950 (MI355X): the sparse-shift
// / quantized forward (1-, 2- and 4-byte elements), the interpolating forward and the backward (fp32, fp16, bf16).
// SURVEY section 8f N3; DESIGN section 3.11.
//
// In NHWC the channels of a pixel are contiguous and every channel has its own shift, so the channel-fastest gather
// of shiftnd_cl.hip sends the 64 lanes of a wave to up to (2 max|shift| + 1)^2 different pixels: one cache-line
// lookup per lane (0.97 TB/s on N16 C256 224x224 fp32).  But every source element is used by a bounded set of output
// elements, so the data can be moved once, through LDS:
//   * a workgroup owns one 128-byte line of channels per pixel (32 / 64 / 128 channels of 4 / 2 / 1 bytes) x a strip
//     of output columns and walks down the rows of a band; LDS holds a ring of source rows with a halo of R = 3 pixels
//     (shifts up to +-3 in each dim take the tiled path per channel, larger ones go element by element at the end);
//   * per step ONE new source row is staged (whole lines, global -> registers three rows ahead -> LDS) and ONE output
//     row is produced; every memory instruction of the row loop is unconditional (raw-buffer addressing drops what must
//     not happen), so the compiler's wait counts are exact and the prefetch really is in flight;
//   * the pixel pitch in LDS is 33 words: lanes that read consecutive channels of arbitrary pixels, or consecutive
//     pixels of one channel, hit different banks -- the gather runs at LDS rate;
//   * forward output: channels-last (the quantized op keeps the format, shifts_quantized.cpp:119-121) or
//     NCHW-contiguous like the reference's float forward (cpu/shifts_cpu.cpp:221): lanes then run along the row of one
//     channel -- the layout change costs nothing extra.
// Periodic padding wraps to the far side of the plane: the pixels near the left / right edge and the rows near the top /
// bottom whose source is not in the ring (at most R of each) are written by the element-by-element pass that also serves
// the shifts beyond the ring; everything else of a periodic call takes the tiled path.
//
// Reference behaviour restated: kernels/shifts_kernels.h:330-400 (nhwdc forward), :402-527 (nhwdc backward), :574-624
// (quantized).  Roofline: HBM, 2 x s bytes per element forward, 3 x s backward.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kR = 3;                 // ring half depth = largest |shift| served from LDS
constexpr int kTW = 32;               // output columns per workgroup
constexpr int kLine = 128;            // bytes of channels per workgroup and pixel: 32 / 64 / 128 channels of 4 / 2 / 1 bytes
constexpr int kPW = kTW + 2 * kR;     // staged pixels per row
constexpr int kRing = 2 * kR + 1;     // staged rows
constexpr int kPitch = kLine / 4 + 1; // words per staged pixel
constexpr int kRowWords = kPW * kPitch;
constexpr int kPieces = kPW * (kLine / 16);   // 16-byte pieces per staged row (304)
constexpr int kNP = (kPieces + kThreads - 1) / kThreads;
#ifndef CLT_DEPTH
#define CLT_DEPTH 3
#endif

struct ClTiledParams {
    const char *x;
    char *out;
    const void *w;
    int64_t wzp;
    uint32_t fill;       // fill element (zero point / 0) in the low bits
    int wkind, N, C, H, W, pad;
    int OH, OW, LH, LW;  // the window: output sizes and its corner in the source image (round 4; no crop: H, W, 0, 0)
    int D, OD, LD;       // ND3 (NDHWC): planes of the source, of the output, the window's first plane (2-D: 1, 1, 0)
    int out_cl;          // output layout: channels-last (1) or NCHW-contiguous (0)
    int wtiles, cblocks, bands, band_rows;
    unsigned xcd_blocks;     // grid / 8 when the XCD-contiguous block remap is on (grid % 8 == 0 and knob 22), else 0
    FastDiv d_wtiles, d_cblocks, d_bands;
    FastDiv d_perH, d_perW, d_perD, d_OD;
};

// integer shifts of NI weights, all loads issued before the first use (gather_shift, one element at a time, pays one
// memory round trip per weight in every workgroup's prologue)
template <int NI>
__device__ __forceinline__ void gather_shifts(const void *w, int wkind, int64_t wzp, const int (&idx)[NI], int64_t (&sh)[NI]) {
#define SHIFTND_CLT_LOAD(TYPE, EXPR) \
    { \
        TYPE raw[NI]; \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) raw[i] = static_cast<const TYPE *>(w)[idx[i]]; \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) { const TYPE r = raw[i]; sh[i] = (EXPR); } \
    } \
    break;
    switch (wkind) {
    case SHIFTND_F32: SHIFTND_CLT_LOAD(float, static_cast<int64_t>(rintf(r)))
    case SHIFTND_F64: SHIFTND_CLT_LOAD(double, static_cast<int64_t>(rint(r)))
    case SHIFTND_F16: SHIFTND_CLT_LOAD(_Float16, static_cast<int64_t>(rintf(static_cast<float>(r))))
    case SHIFTND_BF16: SHIFTND_CLT_LOAD(__bf16, static_cast<int64_t>(rintf(static_cast<float>(r))))
    case SHIFTND_I8: SHIFTND_CLT_LOAD(int8_t, static_cast<int64_t>(r) - wzp)
    case SHIFTND_U8: SHIFTND_CLT_LOAD(uint8_t, static_cast<int64_t>(r) - wzp)
    default: SHIFTND_CLT_LOAD(int32_t, static_cast<int64_t>(r) - wzp)
    }
#undef SHIFTND_CLT_LOAD
}

constexpr uint32_t kOutOfRange = 0x80000000u;   // buffer offset beyond every image (num_records < 2^31): loads give 0, stores are dropped
constexpr int kBufferFlags = 0x00020000;        // raw buffer, 32-bit data format (gfx9 family resource word 3)

template <int ES> struct ElemOf;
template <> struct ElemOf<1> { using type = uint8_t; };
template <> struct ElemOf<2> { using type = uint16_t; };
template <> struct ElemOf<4> { using type = uint32_t; };

// Every memory instruction of the row loop is unconditional and in straight-line code, so that hipcc counts vmcnt
// exactly and the loads issued kDepth rows ahead really stay in flight (a load or store under a thread-dependent
// branch makes it wait for vmcnt(0), i.e. one full memory round trip per row: 0.48 ms instead of 0.33 ms on
// N16 C256 224x224 fp32).  What must not happen is expressed through buffer addressing instead: lanes without a source
// piece / an output use an out-of-range offset, which the hardware answers with zero / drops.
//
// A thread produces 4 output dwords per row; a dword holds NE = 4 / ES elements.  OUT_CL: dword = NE consecutive
// channels of one pixel (thread = (dword of the pixel line 0..31, pixel lane 0..7), pixels pl + 8 i);  otherwise dword
// = NE consecutive columns of one channel row (dword D = thread + 256 i of the tile's 128 / ES channel rows of
// 32 ES bytes).  Either way every element of a thread has a fixed channel and column over the rows.
//
// ND3 (round 4, NDHWC): a workgroup works on ONE output plane dz.  Its ring is filled element by element, every channel from ITS
// source plane fold(dz + LD - shift_d(c)) -- so a ring row holds, for every channel, the row of the plane that channel reads,
// and everything behind the staging is the 2-D kernel unchanged (rows and columns shift through the ring, the depth shift
// through the staging address).  Every source element is read once, whatever the depth shifts are; the price is 4-byte (2-byte)
// loads with the lanes along the channels instead of 16-byte pieces.  4- and 2-byte elements.
template <int ES, bool OUT_CL, bool ND3>
__global__ __launch_bounds__(kThreads) void cl_tiled_forward(const ClTiledParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    using EL = typename ElemOf<ES>::type;
    constexpr int NE = 4 / ES;            // elements per dword
    constexpr int CB = kLine / ES;        // channels per workgroup
    constexpr int RD = 8 * ES;            // NCHW output: dwords per channel-row segment of the tile
    // one dword of fill elements, the ring, dump words for pieces that do not exist.  An element that is not served from the ring
    // (padding, a shift beyond the ring) reads the fill word: its offsets are hugely negative (kNeg) and the address is
    // max(offset sum, 0) -- no lane masks in the row loop (round 4: the loop was scalar-bound, 94 - 220 scalar instructions per row)
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[4 + kRing * kRowWords + 4];
    uint32_t *const ring = ring_all + 4;
    constexpr int kNeg = -(1 << 24);
    if (threadIdx.x == 0) ring_all[0] = ES == 4 ? p.fill : (ES == 2 ? (p.fill & 0xffffu) * 0x10001u : (p.fill & 0xffu) * 0x01010101u);
    __shared__ int tab_sh[CB], tab_sw[CB];             // canonical shifts of the workgroup's channels
    __shared__ int tab_pz[ND3 ? CB : 1];               // ND3: the channel's source plane for this workgroup's output plane (-1: fill)
    constexpr int kDump = kRing * kRowWords;

    // ---- which tile ---------------------------------------------------------------------------------------------
    // XCD-contiguous ids: workgroups that share an XCD (blockIdx % 8) and its L2 own neighbouring tiles (shared halo)
    unsigned b = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    // ND3: the output plane is the FASTEST index -- the workgroups of neighbouring planes read the same pixel lines (each takes the
    // channels whose depth shift points there) and must meet in one L2: with the plane outermost every line came from HBM once per
    // depth shift in use (N8 C128 16x112x112 fp32: 2.87 GB read for a 0.82 GB tensor)
    int dz = 0;
    if constexpr (ND3) {
        const unsigned q = fdiv(b, p.d_OD);
        dz = static_cast<int>(b - q * static_cast<unsigned>(p.OD));
        b = q;
    }
    const int wt = static_cast<int>(b - fdiv(b, p.d_wtiles) * p.wtiles);
    b = fdiv(b, p.d_wtiles);
    const int cb = static_cast<int>(b - fdiv(b, p.d_cblocks) * p.cblocks);
    b = fdiv(b, p.d_cblocks);
    const int band = static_cast<int>(b - fdiv(b, p.d_bands) * p.bands);
    const int n = static_cast<int>(fdiv(b, p.d_bands));
    const int w0 = wt * kTW, c0 = cb * CB;
    // the window (round 4): output rows / columns [0, OH) x [0, OW) read source rows / columns + (LH, LW) through the maps; the
    // ring follows the SOURCE rows hs = h + LH and pixels w0 + LW - R ..
    const int H = p.H, W = p.W, C = p.C, OH = p.OH, OW = p.OW, LH = p.LH, LW = p.LW;
    const int h0 = band * p.band_rows, h1 = min(OH, h0 + p.band_rows);
    const int DZ = ND3 ? p.D : 1, OD = ND3 ? p.OD : 1;
    const char *xn = p.x + static_cast<int64_t>(n) * DZ * H * W * C * ES;
    char *on = p.out + static_cast<int64_t>(n) * OD * OH * OW * C * ES;
    const uint32_t img_bytes = static_cast<uint32_t>(DZ) * static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * static_cast<uint32_t>(C) * ES;  // < 2^31 (host)
    const uint32_t out_bytes = static_cast<uint32_t>(OD) * static_cast<uint32_t>(OH) * static_cast<uint32_t>(OW) * static_cast<uint32_t>(C) * ES;
    const uint32_t plane_bytes = static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * static_cast<uint32_t>(C) * ES;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xn), 0, img_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(on, 0, out_bytes, kBufferFlags);

    // ND3: the channels' shifts FIRST -- the staging addresses depend on the depth shift (one weight round trip before the
    // first row is requested; the 2-D kernel loads its weights while the rows are in flight)
    auto channel_table = [&]() {
        if (threadIdx.x < CB) {
            const int c = min(c0 + static_cast<int>(threadIdx.x), C - 1);
            if constexpr (ND3) {
                const int widx[3] = {3 * c, 3 * c + 1, 3 * c + 2};
                int64_t sh3[3];
                gather_shifts<3>(p.w, p.wkind, p.wzp, widx, sh3);
                tab_pz[threadIdx.x] = DZ == 1 ? 0 : fold_index(dz + p.LD - canon_shift(sh3[0], DZ, p.pad, p.d_perD), DZ, p.pad);   // size-1 dims ignore the shift
                tab_sh[threadIdx.x] = canon_shift(sh3[1], H, p.pad, p.d_perH);
                tab_sw[threadIdx.x] = canon_shift(sh3[2], W, p.pad, p.d_perW);
            } else {
                const int widx[2] = {2 * c, 2 * c + 1};
                int64_t sh2[2];
                gather_shifts<2>(p.w, p.wkind, p.wzp, widx, sh2);
                tab_sh[threadIdx.x] = canon_shift(sh2[0], H, p.pad, p.d_perH);
                tab_sw[threadIdx.x] = canon_shift(sh2[1], W, p.pad, p.d_perW);
            }
        }
        __syncthreads();
    };
    if constexpr (ND3) channel_table();

    // ---- staging: 16-byte pieces of source row y: pixel w0 - R + px, bytes 16 q .. of the channel line ---------------
    // (ND3: elements instead -- element e = px * CB + channel of the staged row, lanes along the channels)
    constexpr int kNS = ND3 ? (kPW * CB + kThreads - 1) / kThreads : kNP;   // loads per thread and staged row
    uint32_t poff[kNS];    // byte offset of the piece / element in row 0 of the image (ND3: of its channel's plane), or out of range
    int pdst[kNS];         // LDS word (ND3: byte) offset within a ring row, or the dump words (-1)
#pragma unroll
    for (int k = 0; k < kNS; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        if constexpr (ND3) {
            const int px = q / CB, ch = q - px * CB, gx = w0 + LW - kR + px;
            const bool elem = px < kPW;
            const int pz = tab_pz[ch];
            poff[k] = (elem && gx >= 0 && gx < W && c0 + ch < C && pz >= 0)
                          ? static_cast<uint32_t>(pz) * plane_bytes + (static_cast<uint32_t>(gx) * C + c0 + ch) * ES : kOutOfRange;
            pdst[k] = elem ? px * (kPitch * 4) + ch * ES : -1;
        } else {
            const int px = q >> 3, cbyte = c0 * ES + (q & 7) * 16, gx = w0 + LW - kR + px;
            const bool piece = q < kPieces;
            poff[k] = (piece && gx >= 0 && gx < W && cbyte < C * ES) ? static_cast<uint32_t>(gx) * C * ES + cbyte : kOutOfRange;
            pdst[k] = piece ? px * kPitch + (q & 7) * 4 : -1;
        }
    }
    const uint32_t row_bytes = static_cast<uint32_t>(W) * C * ES;
    constexpr int kDepth = CLT_DEPTH;  // rows of staging in flight (a workgroup moves only ~5 KB per row)
    using SV = std::conditional_t<ND3, uint32_t, u4>;   // what one staging load returns
    SV pvs[kDepth][kNS];
    auto load_row = [&](int y, int ylast, SV (&pv)[kNS]) {  // rows outside the image or beyond the band: nothing is read
        // (sign arithmetic: selects on "wanted" compiled to branches around duplicated loads)
        const int unwanted = (y >> 31) | ((ylast - y) >> 31);
        const __amdgpu_buffer_rsrc_t r = xres;
        const uint32_t so = static_cast<uint32_t>(y & ~unwanted) * row_bytes, dead = static_cast<uint32_t>(unwanted) & kOutOfRange;
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            if constexpr (!ND3) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(r, poff[k] | dead, so, 0);
            else if constexpr (ES == 4) pv[k] = __builtin_amdgcn_raw_buffer_load_b32(r, poff[k] | dead, so, 0);
            else if constexpr (ES == 2) pv[k] = __builtin_amdgcn_raw_buffer_load_b16(r, poff[k] | dead, so, 0);
            else pv[k] = __builtin_amdgcn_raw_buffer_load_b8(r, poff[k] | dead, so, 0);
        }
    };
    auto store_row = [&](int y, const SV (&pv)[kNS]) {
        const int slot = (y % kRing + kRing) % kRing;
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            if constexpr (ND3) {
                char *d = reinterpret_cast<char *>(ring) + (pdst[k] >= 0 ? slot * (kRowWords * 4) + pdst[k] : kDump * 4);
                *reinterpret_cast<EL *>(d) = static_cast<EL>(pv[k]);
            } else {
                uint32_t *d = ring + (pdst[k] >= 0 ? slot * kRowWords + pdst[k] : kDump);
                d[0] = pv[k].x;
                d[1] = pv[k].y;
                d[2] = pv[k].z;
                d[3] = pv[k].w;
            }
        }
    };
    const int ylast = min(H - 1, h1 - 1 + LH + kR);
    // The kDepth rows for the first steps are requested BEFORE the ring rows: when those have arrived nothing is
    // pending any more, so the loop's wait counts are those of its own back edge (kDepth rows of loads and stores
    // in flight), not the shorter distance of this prologue.
    SV pre[2 * kR][kNS];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_row(h0 + LH + kR + d, ylast, pvs[d]);
#pragma unroll
    for (int r = 0; r < 2 * kR; ++r) load_row(h0 + LH - kR + r, ylast, pre[r]);

    // ---- the channels' shifts (while the rows are in flight) -----------------------------------------------------------
    if constexpr (!ND3) channel_table();

    // ---- thread -> elements -------------------------------------------------------------------------------------------
    const int lane_a = static_cast<int>(threadIdx.x) & 31, lane_b = static_cast<int>(threadIdx.x) >> 5;
    const int perH = map_period(H, p.pad), perW = map_period(W, p.pad);
    // rows of a thread's elements: OUT_CL one canonical row shift per element slot j (channels 4 lane_a / ES + j), else
    // one per dword i (its channel)
    constexpr int NSH = OUT_CL ? NE : 4;
    int csh[NSH], ssh[NSH];    // canonical / signed row shift
    int xoff[4][NE];           // LDS byte offset of the source element within a staged row (kNeg when not served from the ring)
    uint32_t ring_ok = 0, far = 0;   // bit 4 i + j: served from the ring / gathered from memory after the row loop
    uint32_t ooff[4];          // byte offset of the output dword in row h0 of the image, or out of range (nothing to store)
    int gcol[4][NE];           // far: source column
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int D = static_cast<int>(threadIdx.x) + kThreads * i;   // NCHW output: dword of the tile
        bool live_dw = false;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int ch = OUT_CL ? lane_a * NE + j : D / RD;
            const int col = OUT_CL ? lane_b + 8 * i : (D % RD) * NE + j;
            const int c = c0 + ch;
            const bool live = c < C && w0 + col < OW;
            live_dw = live_dw || live;
            const int sh = tab_sh[ch], sw = tab_sw[ch];
            if (OUT_CL) csh[j] = sh;
            else csh[i] = sh;
            int sx = W == 1 ? 0 : fold_index(w0 + LW + min(col, OW - 1 - w0) - sw, W, p.pad);  // size-1 dims ignore the shift
            if constexpr (ND3) sx = tab_pz[ch] < 0 ? -1 : sx;   // the channel's source plane is padding: fill, like a column outside
            gcol[i][j] = sx;
            // canon_shift returns the non-negative representative for the reflecting paddings: look at the signed one
            const int sh_s = (perH && 2 * sh > perH) ? sh - perH : sh, sw_s = (perW && 2 * sw > perW) ? sw - perW : sw;
            // (periodic padding: a column near the edge comes from the far side of the row -- not among the staged pixels)
            const bool in_ring = sh_s >= -kR && sh_s <= kR && sw_s >= -kR && sw_s <= kR && (sx < 0 || (sx >= w0 + LW - kR && sx < w0 + LW + kTW + kR));
            const bool nr = live && sx >= 0 && in_ring;
            if (OUT_CL) ssh[j] = sh_s;
            else ssh[i] = sh_s;
            ring_ok |= (nr ? 1u : 0u) << (4 * i + j);
            far |= ((live && sx >= 0 && !in_ring) ? 1u : 0u) << (4 * i + j);
            xoff[i][j] = nr ? (sx - (w0 + LW - kR)) * (kPitch * 4) + ch * ES : kNeg;
        }
        // (C * ES and W * ES are multiples of 4 where it matters: a dword is live or dead as a whole)
        const int ch0 = OUT_CL ? lane_a * NE : D / RD, col0 = OUT_CL ? lane_b + 8 * i : (D % RD) * NE;
        const uint32_t o = OUT_CL ? (static_cast<uint32_t>((dz * OH + h0) * OW + w0 + col0) * C + c0 + ch0) * ES
                                  : (static_cast<uint32_t>(((c0 + ch0) * OD + dz) * OH + h0) * OW + w0 + col0) * ES;
        ooff[i] = live_dw ? o : kOutOfRange;
    }
    const uint32_t ostep = static_cast<uint32_t>(OUT_CL ? OW * C : OW) * ES;

    // rows h0 - R .. h0 + R - 1 of the ring
#pragma unroll
    for (int r = 0; r < 2 * kR; ++r) {
        const int y = h0 + LH - kR + r;
        if (y >= 0 && y < H) store_row(y, pre[r]);
    }
    const uint8_t *ringz = reinterpret_cast<const uint8_t *>(ring_all);   // byte 0: the fill word; the ring starts at byte 16
    const bool periodic = p.pad == 2 && H > 1;
    // source rows of shifts within the ring with ONE fold, in bit arithmetic (cl_tiled_backward): valid for H > R or H == 1
    // (host: smaller images keep the channel-fastest kernels).  Periodic: a row that wraps is not in the ring -- negative like the zero padding (the fill
    // value goes out and the pass after the loop writes the element)
    const int fm = (p.pad == 3 || p.pad == 4) ? -1 : 0;
    const int fLo = p.pad == 4 ? -1 : 0, fHi = p.pad == 1 ? H - 1 : (p.pad == 3 ? 2 * H - 2 : 2 * H - 1);
    const int zneg = (p.pad == 0 || p.pad == 2) ? kNeg : 0;
    auto fold_once = [&](int idx) {
        const int below = idx >> 31, above = (H - 1 - idx) >> 31, t = idx & fm;
        const int r = (idx & ~(below | above)) | ((fLo - t) & below) | ((fHi - t) & above) | ((below | above) & zneg);
        return H == 1 ? 0 : r;
    };
    auto step = [&](int h, SV (&pv)[kNS]) {
        const int hs = h + LH;   // the source row of output row h under a zero shift
        __syncthreads();  // everybody is done with the slot that row hs + R replaces (row hs - R - 1)
        if (hs + kR < H) store_row(hs + kR, pv);
        __syncthreads();
        load_row(hs + kDepth + kR, ylast, pv);  // in flight while this and the next rows are produced
        const uint32_t so = static_cast<uint32_t>(h - h0) * ostep;
        int rowb[NSH];        // LDS byte offset (from ringz) of the source row of each shift, or negative (padding)
#pragma unroll
        for (int k = 0; k < NSH; ++k) {
            const int sy = fold_once(hs - ssh[k]);
            const uint32_t r = static_cast<uint32_t>(sy & ~(sy >> 31));    // (0 for a negative row)
            const uint32_t sl = r - __umulhi(r, 613566757u) * kRing;       // r % 7 (r < 2^20)
            rowb[k] = (16 + static_cast<int>(sl) * (kRowWords * 4)) | ((sy >> 31) & kNeg);
        }
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = 0;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const uint32_t e = *reinterpret_cast<const EL *>(ringz + max(rowb[OUT_CL ? j : i] + xoff[i][j], 0));
                v[i] |= e << (8 * ES * j);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(v[i], ores, ooff[i], so, 0);
    };
    int hb = h0;
    for (; hb + kDepth <= h1; hb += kDepth) {   // whole groups: no condition between the steps (exact wait counts)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) step(hb + d, pvs[d]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)
        if (hb + d < h1) step(hb + d, pvs[d]);

    // ---- shifts beyond the ring: gathered from memory, element by element (rare).  The row loop stored the fill
    // value in their place; those stores are complete before the elements are written again. ------------------------
    const bool wrap_rows = periodic && (h0 + LH < kR || h1 + LH > H - kR);   // the band holds rows whose source row wraps
    if (far || (wrap_rows && ring_ok)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const bool all_rows = (far >> (4 * i + j)) & 1u;
                if (!all_rows && !(wrap_rows && ((ring_ok >> (4 * i + j)) & 1u))) continue;
                const int D = static_cast<int>(threadIdx.x) + kThreads * i;
                const int ch = OUT_CL ? lane_a * NE + j : D / RD;
                const int col = OUT_CL ? lane_b + 8 * i : (D % RD) * NE + j;
                const EL *xe = reinterpret_cast<const EL *>(xn);
                EL *o = reinterpret_cast<EL *>(on) + (OUT_CL ? ((static_cast<int64_t>(dz) * OH + h0) * OW + w0 + col) * C + c0 + ch
                                                             : ((static_cast<int64_t>(c0 + ch) * OD + dz) * OH + h0) * OW + w0 + col);
                const int64_t zoff = ND3 ? static_cast<int64_t>(tab_pz[ch]) * H * W * C : 0;   // the channel's source plane (far: pz >= 0)
                const int shc = csh[OUT_CL ? j : i], shs = ssh[OUT_CL ? j : i];
                for (int h = h0; h < h1; ++h) {
                    const int sy = H == 1 ? 0 : fold_index(h + LH - shc, H, p.pad);
                    // (an element of the ring, periodic padding: exactly the rows the loop took for padding -- the source row
                    //  under the signed shift lies outside the image)
                    if (all_rows || h + LH - shs < 0 || h + LH - shs >= H)
                        *o = sy >= 0 ? xe[zoff + (static_cast<int64_t>(sy) * W + gcol[i][j]) * C + c0 + ch] : static_cast<EL>(p.fill);
                    o += OUT_CL ? OW * C : OW;
                }
            }
        }
    }
}

// =====================================================================================================
// Backward for dense channels-last fp32 tensors (saved input, incoming gradient and grad_x all NHWC): the same
// row walk with TWO rings, rows h - R .. h + R + 1 of the input and of the incoming gradient (the + 1: the
// interpolation corners, which the sparse shift's weight gradient reads too, shifts_kernels.h:271-283), 16 output
// columns x 32 channels per workgroup (2 x 24 KB of LDS: three workgroups per CU).  A thread keeps ONE channel
// (lane % 32) and two pixels per row, so the weight-gradient sums stay in its registers (fp64, like every backward
// kernel here) and are combined once per workgroup, in a fixed order, into the [group][C][3] partial sums that
// reduce_weight_grads finishes.  grad_x: the sparse shift reads grad_out at o + shift (one tap, raw copy), the active
// shift interpolates the four corners around o - floor(shift) (:287-293).  Reference: kernels/shifts_kernels.h:402-527.
// Roofline: HBM, 3 x 4 bytes per element.
// =====================================================================================================
constexpr int kBTW = 16;                       // output columns per workgroup
constexpr int kBPW = kBTW + 2 * kR + 1;        // staged pixels per row
constexpr int kBRing = 8;                      // staged rows h - R .. h + R + 1
constexpr int kBPieces = kBPW * (kLine / 16);  // 16-byte pieces per staged row and tensor (184: one per thread)
static_assert(kBPieces <= kThreads, "one piece of each tensor per thread");

struct ClTiledBwdParams {
    const char *x, *go;
    char *gx;
    const void *w;
    double *partials;    // [N * bands * wtiles][C][3]
    int wkind, N, C, H, W, pad;
    int OH, OW, LH, LW;  // the window (round 4): grad_out's sizes and its corner in the input image (no crop: H, W, 0, 0)
    int go_nchw;         // the incoming gradient is NCHW-contiguous (saved input and grad_x: channels-last)
    int go_pieces;       // ... and its rows are whole, aligned 16-byte pieces: staged by 16-byte loads (4-byte elements)
    int wtiles, cblocks, bands, band_rows;
    unsigned xcd_blocks;     // grid / 8 when the XCD-contiguous block remap is on (grid % 8 == 0 and knob 22), else 0
    FastDiv d_wtiles, d_cblocks, d_bands;
    FastDiv d_perH, d_perW, d_perOH, d_perOW;
};

// GO_NCHW: the incoming gradient is NCHW-contiguous (what the op downstream of the reference's float forward hands back:
// that forward returns an NCHW tensor even for a channels-last input, cpu/shifts_cpu.cpp:221) while the saved input and
// grad_x are channels-last.  The gradient ring is filled from channel rows instead of pixel lines: CB segments of kBPW
// consecutive elements per staged row, element loads with lanes along the pixels (a wave touches ~3 runs of 92 / 46
// bytes), written to the same [pixel][channel] ring (pitch 33 words: lanes along the pixels of one channel hit
// different banks).  Everything after the staging is the same kernel.
// TW (round 6): grad_x columns per workgroup.  16: one staged piece of each tensor per thread, 2 x 24 KB of LDS, three workgroups per CU;
// 32 (channels-last gradient only): a halo of 39 / 32 staged pixels instead of 23 / 16 (the loads' amplification 1.22 x instead of
// 1.44 x), twice the work between two barriers, two pieces per tensor and thread, 80 KB of LDS (two workgroups per CU).
template <typename T, bool ACTIVE, bool GO_NCHW, int TW = kBTW>
__global__ __launch_bounds__(kThreads) void cl_tiled_backward(const ClTiledBwdParams p) {
    static_assert(TW == kBTW || !GO_NCHW, "the wide strip takes a channels-last gradient");
    constexpr int BPW = TW + 2 * kR + 1;             // staged pixels per row
    constexpr int BPIECES = BPW * (kLine / 16);      // 16-byte pieces per staged row and tensor
    constexpr int NPX = (BPIECES + kThreads - 1) / kThreads;   // ... per thread (1 or 2)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    using S = typename T::S;
    using CT = typename T::C;
    static_assert(sizeof(S) == 4 || sizeof(S) == 2, "fp32, fp16, bf16");
    constexpr int ES = sizeof(S), CB = kLine / ES;   // channels per workgroup: 32 (fp32) or 64 (16-bit types)
    constexpr int PL = kThreads / CB, NI = TW / PL;   // pixel lanes; pixels per thread and row (2 or 4)
    // four zero words, input ring, gradient ring, dump words.  The zero words are what a padding tap reads: row and column offsets
    // of taps that do not exist are hugely negative (kNeg), the address is max(offset sum, 0) -- no lane masks, no selects, no
    // exec-mask regions in the row loop (round 4: the loop issued 137 scalar instructions per row, and the CU's one scalar unit
    // was busy for two thirds of the kernel's time)
    // words per staged pixel: 32 -- every LDS read of the row loop has its lanes along the CHANNELS of (per lane different) pixels,
    // bank = channel word whatever the pixel; the padded pitch of the forward kernels (33: their NCHW-output form reads along the
    // pixels) made half of this kernel's LDS cycles bank conflicts.  GO_NCHW stages along the pixels and keeps 33.
    // (the INPUT ring is staged by pieces and read along the channels in every form: 32; only the gradient ring of GO_NCHW is 33)
    constexpr int PITCHX = kLine / 4, PITCHG = GO_NCHW ? kPitch : kLine / 4;
    constexpr int BROWX = BPW * PITCHX, BROWG = BPW * PITCHG, BRINGX = kBRing * BROWX, BRINGG = kBRing * BROWG;
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[4 + BRINGX + BRINGG + 4];
    uint32_t *const ring = ring_all + 4;
    constexpr int kDump = BRINGX + BRINGG;
    constexpr int kNeg = -(1 << 24);
    if (threadIdx.x < 4) ring_all[threadIdx.x] = 0u;   // (read after the first barrier of the row loop)

    // ---- which tile ---------------------------------------------------------------------------------------------
    // XCD-contiguous ids: workgroups that share an XCD (blockIdx % 8) and its L2 own neighbouring tiles (shared halo)
    unsigned b = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int cb = static_cast<int>(b - fdiv(b, p.d_cblocks) * p.cblocks);   // channel blocks of one pixel tile are neighbours
    b = fdiv(b, p.d_cblocks);
    const int pidx = static_cast<int>(b);                                     // (n, band, wt): the partial-sum group
    const int wt = static_cast<int>(b - fdiv(b, p.d_wtiles) * p.wtiles);
    b = fdiv(b, p.d_wtiles);
    const int band = static_cast<int>(b - fdiv(b, p.d_bands) * p.bands);
    const int n = static_cast<int>(fdiv(b, p.d_bands));
    const int w0 = wt * TW, c0 = cb * CB;
    const int h0 = band * p.band_rows, h1 = min(p.H, h0 + p.band_rows);
    // The window (round 4; shifts_kernels.h:402-527 with the borders of shifts.cpp:93-135): grad_out has the window's sizes OH x OW
    // and sits at (LH, LW) of the input image.  The kernel stays in INPUT coordinates: the gradient ring's row y / pixel gxs hold
    // grad_out's row y - LH / column gxs - LW (nothing outside the window), the gradient maps fold in the window's sizes and are
    // moved back by (LH, LW); input elements outside the window get a zero gradient and add nothing to the weight gradients.
    const int H = p.H, W = p.W, C = p.C, OH = p.OH, OW = p.OW, LH = p.LH, LW = p.LW;
    const int64_t img = static_cast<int64_t>(n) * H * W * C * ES;
    const char *xn = p.x + img, *gn = p.go + static_cast<int64_t>(n) * OH * OW * C * ES;
    char *on = p.gx + img;
    const uint32_t img_bytes = static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * static_cast<uint32_t>(C) * ES;  // < 2^31 (host)
    const uint32_t go_bytes = static_cast<uint32_t>(OH) * static_cast<uint32_t>(OW) * static_cast<uint32_t>(C) * ES;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xn), 0, img_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gn), 0, go_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(on, 0, img_bytes, kBufferFlags);
    // the gradient element (row r, column cc: grad_out's own coordinates) of channel ch, in elements from the image base
    auto g_index = [&](int ch, int r, int cc) {
        return GO_NCHW ? (static_cast<int64_t>(ch) * OH + r) * OW + cc : (static_cast<int64_t>(r) * OW + cc) * C + ch;
    };

    // ---- staging: one 16-byte piece of the input row and one of the gradient row per thread ------------------------------
    const int q = static_cast<int>(threadIdx.x);
    uint32_t poff[NPX], poffg[NPX];
    int pdst[NPX];   // (also the gradient's when it is channels-last: PITCHG == PITCHX then)
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int qk = q + k * kThreads;
        const bool piece = qk < BPIECES;
        const int px = qk >> 3, part = qk & 7, gxs = w0 - kR + px;
        poff[k] = (piece && gxs >= 0 && gxs < W && c0 * ES + part * 16 < C * ES) ? static_cast<uint32_t>(gxs) * C * ES + c0 * ES + part * 16 : kOutOfRange;
        poffg[k] = (piece && gxs >= LW && gxs < LW + OW && c0 * ES + part * 16 < C * ES) ? static_cast<uint32_t>(gxs - LW) * C * ES + c0 * ES + part * 16 : kOutOfRange;
        pdst[k] = piece ? px * PITCHX + part * 4 : -1;
    }
    const uint32_t row_bytes = static_cast<uint32_t>(W) * C * ES;
    constexpr int kDepth = CLT_DEPTH;
    // GO_NCHW: element e = ch * BPW + px of the staged gradient row, kGN per thread
    constexpr int kGE = CB * BPW, kGN = GO_NCHW ? (kGE + kThreads - 1) / kThreads : 1;
    struct GRow {
        u4 v[NPX];           // the thread's 16-byte pieces of a channels-last gradient row
        uint32_t e[kGN];     // GO_NCHW: its elements of the row's channel segments
    };
    uint32_t goff[kGN];      // byte offset of the element in row 0 of the (NCHW) image, or out of range
    int gdst[kGN];           // LDS byte offset within a ring row, or -1 (dump)
    if constexpr (GO_NCHW) {
#pragma unroll
        for (int k = 0; k < kGN; ++k) {
            const int e = k * kThreads + q, ch = e / BPW, pxe = e - ch * BPW, gxe = w0 - kR + pxe;
            const bool ok = e < kGE && gxe >= LW && gxe < LW + OW && c0 + ch < C;
            goff[k] = ok ? (static_cast<uint32_t>(c0 + ch) * OH * OW + (gxe - LW)) * ES : kOutOfRange;
            gdst[k] = e < kGE ? pxe * (PITCHG * 4) + ch * ES : -1;
        }
    }
    // GO_NCHW, 4-byte elements, gradient rows of whole 16-byte pieces (host: p.go_pieces): the channel segments by 16-byte loads --
    // the 23 staged pixels of a channel lie in at most 7 aligned pieces, 32 channels x 7 = 224 pieces: ONE load per thread and row
    // instead of three element loads; the four elements of a piece go to four pixels of the ring (or to the dump word).
    constexpr int kGPC = 7;                      // pieces per channel segment
    const bool g_pieces = GO_NCHW && ES == 4 && p.go_pieces != 0;   // (launch-uniform)
    uint32_t gpoff = kOutOfRange;
    int gpdst[4] = {-1, -1, -1, -1};
    if constexpr (GO_NCHW && ES == 4) {
        const int ch = q / kGPC, pi = q - ch * kGPC;
        const int cstart = w0 - kR - LW;                       // grad_out column of ring pixel 0
        const int e0 = ((cstart >> 2) + pi) * 4;               // ... of this piece's first element (floor: cstart may be negative)
        const bool ok = q < CB * kGPC && e0 >= 0 && e0 < OW && c0 + ch < C;
        gpoff = ok ? (static_cast<uint32_t>(c0 + ch) * OH * OW + e0) * ES : kOutOfRange;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pxj = e0 + j - cstart;
            gpdst[j] = (q < CB * kGPC && pxj >= 0 && pxj < BPW) ? pxj * (PITCHG * 4) + ch * ES : -1;
        }
    }
    const uint32_t grow_bytes = static_cast<uint32_t>(OW) * (GO_NCHW ? 1 : C) * ES;
    struct XRow {
        u4 v[NPX];
    };
    XRow pvx[kDepth];
    GRow pvg[kDepth];
    auto load_row = [&](int y, int ylast, XRow &vx, GRow &vg) {  // rows outside the image or beyond the band: nothing is read
        // (a row that is not wanted: the out-of-range bit in the vector offset -- one scalar select per row; choosing between the
        //  image's resource and an empty one compiled to branches around duplicated loads)
        //  image's resource and an empty one compiled to branches around duplicated loads, and so did selects on "wanted": sign
        //  arithmetic instead)
        const int unwanted = (y >> 31) | ((ylast - y) >> 31);   // -1: above the image or beyond what the band needs
        const uint32_t so = static_cast<uint32_t>(y & ~unwanted) * row_bytes, dead = static_cast<uint32_t>(unwanted) & kOutOfRange;
#pragma unroll
        for (int k = 0; k < NPX; ++k) vx.v[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, poff[k] | dead, so, 0);
        const int unwanted_g = unwanted | ((y - LH) >> 31) | ((LH + OH - 1 - y) >> 31);   // grad_out's row y - LH
        const uint32_t sg = static_cast<uint32_t>((y - LH) & ~unwanted_g) * grow_bytes, dead_g = static_cast<uint32_t>(unwanted_g) & kOutOfRange;
        if constexpr (GO_NCHW) {
            if (g_pieces) {
                vg.v[0] = __builtin_amdgcn_raw_buffer_load_b128(gres, gpoff | dead_g, sg, 0);
            } else {
#pragma unroll
                for (int k = 0; k < kGN; ++k) {
                    if constexpr (ES == 4) vg.e[k] = __builtin_amdgcn_raw_buffer_load_b32(gres, goff[k] | dead_g, sg, 0);
                    else vg.e[k] = __builtin_amdgcn_raw_buffer_load_b16(gres, goff[k] | dead_g, sg, 0);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPX; ++k) vg.v[k] = __builtin_amdgcn_raw_buffer_load_b128(gres, poffg[k] | dead_g, sg, 0);
        }
    };
    auto store_row = [&](int y, const XRow &vx, const GRow &vg) {
        const int slot = y & (kBRing - 1);
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            uint32_t *dx = ring + (pdst[k] >= 0 ? slot * BROWX + pdst[k] : kDump);
            *reinterpret_cast<u4 *>(__builtin_assume_aligned(dx, 16)) = vx.v[k];   // 16-byte aligned pieces: one ds_write_b128, the 8 pieces of a pixel cover all banks
        }
        if constexpr (GO_NCHW) {
            char *gbase = reinterpret_cast<char *>(ring) + (BRINGX + slot * BROWG) * 4;
            char *dump = reinterpret_cast<char *>(ring) + kDump * 4;
            if (g_pieces) {
                const uint32_t ge[4] = {vg.v[0].x, vg.v[0].y, vg.v[0].z, vg.v[0].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t *>(gpdst[j] >= 0 ? gbase + gpdst[j] : dump) = ge[j];
            } else {
#pragma unroll
                for (int k = 0; k < kGN; ++k) {
                    char *d = gdst[k] >= 0 ? gbase + gdst[k] : dump;
                    if constexpr (ES == 4) *reinterpret_cast<uint32_t *>(d) = vg.e[k];
                    else *reinterpret_cast<uint16_t *>(d) = static_cast<uint16_t>(vg.e[k]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
                uint32_t *dg = ring + (pdst[k] >= 0 ? BRINGX + slot * BROWG + pdst[k] : kDump);
                *reinterpret_cast<u4 *>(__builtin_assume_aligned(dg, 16)) = vg.v[k];
            }
        }
    };
    const int ylast = min(H - 1, h1 + kR);
    XRow prex[2 * kR + 1];
    GRow preg[2 * kR + 1];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_row(h0 + kR + 1 + d, ylast, pvx[d], pvg[d]);
#pragma unroll
    for (int r = 0; r <= 2 * kR; ++r) load_row(h0 - kR + r, ylast, prex[r], preg[r]);

    // ---- the thread's channel and its two pixels ---------------------------------------------------------------------
    const int lane_a = static_cast<int>(threadIdx.x) % CB, lane_b = static_cast<int>(threadIdx.x) / CB;
    const int c = c0 + lane_a;
    const bool live_c = c < C;
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    {
        const int wcol[3] = {-1, 0, 1};
        CT wv[3];
        load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(live_c ? c : C - 1) * 2, wcol, wv);
        prep_shift_backward<CT>(wv[1], ACTIVE, sh[1], dw[0]);
        prep_shift_backward<CT>(wv[2], ACTIVE, sh[2], dw[1]);
    }
    const int csxH = canon_shift(sh[1], H, p.pad, p.d_perH), csxW = canon_shift(sh[2], W, p.pad, p.d_perW);
    // grad_x source: the sparse shift reads grad_out at o + shift, the active one at o - shift (shifts_kernels.h:287-293)
    // (the gradient's maps fold in the window's sizes: their own canonical shifts)
    const int csgH = canon_shift(ACTIVE ? sh[1] : -sh[1], OH, p.pad, p.d_perOH), csgW = canon_shift(ACTIVE ? sh[2] : -sh[2], OW, p.pad, p.d_perOW);
    const int perH = map_period(H, p.pad), perW = map_period(W, p.pad), perOH = map_period(OH, p.pad), perOW = map_period(OW, p.pad);
    const int sh_s = (perH && 2 * csxH > perH) ? csxH - perH : csxH, sw_s = (perW && 2 * csxW > perW) ? csxW - perW : csxW;
    const int gh_s = (perOH && 2 * csgH > perOH) ? csgH - perOH : csgH, gw_s = (perOW && 2 * csgW > perOW) ? csgW - perOW : csgW;
    const bool in_ring = sh_s >= -kR && sh_s <= kR && sw_s >= -kR && sw_s <= kR && gh_s >= -kR && gh_s <= kR && gw_s >= -kR && gw_s <= kR;
    const bool near_c = live_c && in_ring, far_c = live_c && !in_ring;
    auto fold_h = [&](int idx) { return H == 1 ? 0 : fold_index(idx, H, p.pad); };   // size-1 dims ignore the shift
    auto fold_w = [&](int idx) { return W == 1 ? 0 : fold_index(idx, W, p.pad); };
    // the gradient's maps: grad_out coordinates in, INPUT coordinates out (-1: padding)
    auto fold_gh = [&](int idx) { const int r = OH == 1 ? 0 : fold_index(idx, OH, p.pad); return r < 0 ? -1 : r + LH; };
    auto fold_gw = [&](int idx) { const int r = OW == 1 ? 0 : fold_index(idx, OW, p.pad); return r < 0 ? -1 : r + LW; };
    // per pixel: LDS byte offsets (within a ring row) of the input corners' columns and of the gradient taps' columns; kNeg: padding.
    // A reflected corner can land one step outside the rings (reflect padding, last column / row, shift -R: the
    // corner at distance R + 1 comes back at distance -(R + 1)); such pixels (`scol`) and rows (`skip` below) are left
    // to the element-by-element pass at the end.
    int xc0[NI], xc1[NI], gc0[NI], gc1[NI], gd[NI];
    uint32_t ooff[NI];
    bool live[NI], scol[NI], cpass[NI];   // cpass: the pixel's column lies in the window
    int lcount[NI];                       // -1: live and in the window (its terms count), else 0
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int col = lane_b + PL * i, wq = w0 + min(col, W - 1 - w0);
        cpass[i] = wq >= LW && wq < LW + OW;
        // (a column outside the window: zero gradient, nothing counted -- every map "padding")
        const int a0 = cpass[i] ? fold_w(wq - csxW) : -1, a1 = cpass[i] ? fold_w(wq - csxW + 1) : -1;
        const int b0 = cpass[i] ? fold_gw(wq - LW - csgW) : -1, b1 = (ACTIVE && cpass[i]) ? fold_gw(wq - LW - csgW + 1) : -1;
        auto outside = [&](int sx) { return sx >= 0 && (sx < w0 - kR || sx > w0 + TW + kR); };
        scol[i] = near_c && w0 + col < W && (outside(a0) || outside(a1) || outside(b0) || outside(b1));
        live[i] = near_c && w0 + col < W && !scol[i];
        lcount[i] = (live[i] && cpass[i]) ? -1 : 0;
        auto lds_col = [&](int sx, int pitch) { return (live[i] && sx >= 0) ? (sx - (w0 - kR)) * (pitch * 4) + lane_a * ES : kNeg; };
        xc0[i] = lds_col(a0, PITCHX);
        xc1[i] = lds_col(a1, PITCHX);
        gc0[i] = lds_col(b0, PITCHG);
        gc1[i] = lds_col(b1, PITCHG);
        gd[i] = (col + kR) * (PITCHG * 4) + lane_a * ES;
        ooff[i] = live[i] ? (static_cast<uint32_t>(h0 * W + w0 + col) * C + c) * ES : kOutOfRange;
    }
    const uint32_t ostep = static_cast<uint32_t>(W) * C * ES;

#pragma unroll
    for (int r = 0; r <= 2 * kR; ++r) {
        const int y = h0 - kR + r;
        if (y >= 0 && y < H) store_row(y, prex[r], preg[r]);
    }
    const char *ringz = reinterpret_cast<const char *>(ring_all);   // byte 0: the zero words
    constexpr int ringx = 16, ringg = 16 + BRINGX * 4;         // the rings, bytes from ringz
    double acc[2] = {0.0, 0.0};
    auto lds_s = [&](int base, int row, int colo) {   // a staged element (storage type); padding (a negative offset) reads a zero word
        return *reinterpret_cast<const S *>(ringz + max(base + row + colo, 0));
    };
    auto lds_f = [&](int base, int row, int colo) { return widen<T>(lds_s(base, row, colo)); };
    auto row_off = [&](int sy) { return ((sy & (kBRing - 1)) * (BROWX * 4)) | ((sy >> 31) & kNeg); };   // (a negative row: padding)
    auto row_off_g = [&](int sy) { return ((sy & (kBRing - 1)) * (BROWG * 4)) | ((sy >> 31) & kNeg); };
    // Source rows with ONE fold of the signed shift (|shift| <= R and H >= 5, or H == 1: the host routes nothing else
    // here): idx in [h - R, h + R + 1] leaves [0, H) by at most R + 1 on one side.  Launch-uniform coefficients instead of
    // fold_index's switch: r = idx inside, aLo - m idx below, aHi - m idx above (border: m = 0; reflect / symmetric:
    // m = 1), -1 outside for the zero padding.
    // Bit arithmetic, not selects: the nested selects compiled to exec-mask regions (six per row).
    const int fm = (p.pad == 3 || p.pad == 4) ? -1 : 0;
    const int fLo = p.pad == 4 ? -1 : 0, fHi = p.pad == 1 ? H - 1 : (p.pad == 3 ? 2 * H - 2 : 2 * H - 1);
    const int zneg = p.pad == 0 ? kNeg : 0;   // zeros padding: outside -> negative
    auto fold_once = [&](int idx, int n, int lo_c, int hi_c) {
        const int below = idx >> 31, above = (n - 1 - idx) >> 31, t = idx & fm;
        const int r = (idx & ~(below | above)) | ((lo_c - t) & below) | ((hi_c - t) & above) | ((below | above) & zneg);
        return n == 1 ? 0 : r;   // (size-1 dims ignore the shift; launch-uniform)
    };
    auto fold1 = [&](int idx) { return fold_once(idx, H, fLo, fHi); };
    // the gradient's rows, folded once in the window's sizes (OH >= 5 or OH == 1: host), grad_out row in, input row out
    const int gLo = p.pad == 4 ? -1 : 0, gHi = p.pad == 1 ? OH - 1 : (p.pad == 3 ? 2 * OH - 2 : 2 * OH - 1);
    auto fold1g = [&](int idx) { return fold_once(idx, OH, gLo, gHi) + LH; };   // (padding stays negative: LH < 2^20)
    // the one source row the rings cannot hold: reflect padding, last row (of the image: the input's corners; of the window: the
    // active shift's gradient taps), shift -R (its + 1 corner, at distance R + 1, comes back at distance -(R + 1))
    const bool srow = near_c && p.pad == 3 && H > 1 && sh_s == -kR && h1 == H;
    const bool srow_g = ACTIVE && near_c && p.pad == 3 && OH > 1 && gh_s == -kR;
    // periodic padding: rows whose input corners or gradient taps wrap to the far side of the image / the window (at most R + 1
    // at the top or the bottom) are not in the rings: left to the element-by-element pass
    const bool periodic = p.pad == 2 && H > 1;
    auto wraps = [&](int h) {
        const int g0 = h - LH - gh_s;   // the gradient taps' first row, grad_out coordinates (the active shift's second: + 1)
        const bool in_rows = h >= LH && h < LH + OH;
        return periodic && near_c && (h - sh_s < 0 || h + 1 - sh_s >= H || (in_rows && OH > 1 && (g0 < 0 || g0 + (ACTIVE ? 1 : 0) >= OH)));
    };
    int xrow1 = fold1(h0 - sh_s);   // the + 1 corner of step h - 1 is the first corner of step h
    int grow1 = fold1g(h0 - LH - gh_s);   // ... and so is the active shift's second gradient row
    auto step = [&](int h, XRow &vx, GRow &vg) {
        __syncthreads();  // everybody is done with the slot that row h + R + 1 replaces (row h - R - 1)
        if (h + kR + 1 < H) store_row(h + kR + 1, vx, vg);
        __syncthreads();
        load_row(h + kDepth + kR + 1, ylast, vx, vg);
        const uint32_t so = static_cast<uint32_t>(h - h0) * ostep;
        const int xr0 = row_off(xrow1);
        xrow1 = fold1(h + 1 - sh_s);
        const int xr1 = row_off(xrow1);
        const bool rpass = h >= LH && h < LH + OH;   // the row lies in the window (else: zero gradient, nothing counted)
        int gr0, gr1 = kNeg;
        if constexpr (ACTIVE) {
            gr0 = rpass ? row_off_g(grow1) : kNeg;
            grow1 = fold1g(h + 1 - LH - gh_s);
            gr1 = rpass ? row_off_g(grow1) : kNeg;
        } else {
            gr0 = rpass ? row_off_g(fold1g(h - LH - gh_s)) : kNeg;
        }
        const int gdr = (h & (kBRing - 1)) * (BROWG * 4);
        int skip = 0;   // -1: this thread leaves the row to the element pass (launch-uniform guards: one padding mode each)
        if (p.pad == 3) skip = ((srow && h == H - 1) || (srow_g && h == LH + OH - 1)) ? -1 : 0;
        else if (p.pad == 2) skip = wraps(h) ? -1 : 0;
        const int stepok = (rpass ? -1 : 0) & ~skip;
        S res[NI];
        CT s0 = CT(0), s1 = CT(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            CT v[4], wg[3];
            v[0] = lds_f(ringx, xr0, xc0[i]);
            v[1] = lds_f(ringx, xr1, xc0[i]);
            v[2] = lds_f(ringx, xr0, xc1[i]);
            v[3] = lds_f(ringx, xr1, xc1[i]);
            const CT graw = widen<T>(*reinterpret_cast<const S *>(ringz + ringg + gdr + gd[i]));
            const uint32_t counted = static_cast<uint32_t>(lcount[i] & stepok);
            weight_grads_nd<2, CT>(v, dw, wg);
            // (masked bit by bit, not multiplied by zero: the corners of a skipped row can come from ring slots that were never
            // staged -- periodic padding, a source row past the image -- and 0 * NaN is not 0; a mask of lanes would be scalar work)
            static_assert(sizeof(CT) == 4, "fp32 compute type");
            s0 += __builtin_bit_cast(CT, __builtin_bit_cast(uint32_t, graw * wg[0]) & counted);
            s1 += __builtin_bit_cast(CT, __builtin_bit_cast(uint32_t, graw * wg[1]) & counted);
            if constexpr (ACTIVE) {
                v[0] = lds_f(ringg, gr0, gc0[i]);
                v[1] = lds_f(ringg, gr1, gc0[i]);
                v[2] = lds_f(ringg, gr0, gc1[i]);
                v[3] = lds_f(ringg, gr1, gc1[i]);
                res[i] = narrow<T>(interp_t<T, 2>(v, dw));
            } else {
                res[i] = lds_s(ringg, gr0, gc0[i]);   // pure copy: the bit pattern is kept
            }
        }
        acc[0] += static_cast<double>(s0);   // (the pixels' terms of one row are added in fp32 first)
        acc[1] += static_cast<double>(s1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (ES == 4) {
                uint32_t bits;
                __builtin_memcpy(&bits, &res[i], 4);
                __builtin_amdgcn_raw_buffer_store_b32(bits, ores, ooff[i] | (static_cast<uint32_t>(skip) & kOutOfRange), so, 0);
            } else {
                uint16_t bits;
                __builtin_memcpy(&bits, &res[i], 2);
                __builtin_amdgcn_raw_buffer_store_b16(bits, ores, ooff[i] | (static_cast<uint32_t>(skip) & kOutOfRange), so, 0);
            }
        }
    };
    int hb = h0;
    for (; hb + kDepth <= h1; hb += kDepth) {   // whole groups: no condition between the steps (exact wait counts)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) step(hb + d, pvx[d], pvg[d]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)
        if (hb + d < h1) step(hb + d, pvx[d], pvg[d]);

    // ---- what the rings could not serve, from memory, element by element.  One element of channel `ch` (canonical shifts cxH .. cgW,
    // fractions fr) at row h, column wq of the image: grad_x written, the weight-gradient terms added to (t0, t1) --------------------
    auto far_element = [&](int ch, int h, int wq, int cxH, int cxW, int cgH, int cgW, const CT (&fr)[3], double &t0, double &t1) {
        const S *xe = reinterpret_cast<const S *>(xn) + ch, *ge = reinterpret_cast<const S *>(gn);
        S *oe = reinterpret_cast<S *>(on) + ch;
        auto tap_s = [&](const S *base, int r, int cc) { return (r >= 0 && cc >= 0) ? base[(static_cast<int64_t>(r) * W + cc) * C] : narrow<T>(CT(0)); };
        auto tap = [&](const S *base, int r, int cc) { return widen<T>(tap_s(base, r, cc)); };
        auto gtap_s = [&](int r, int cc) { return (r >= 0 && cc >= 0) ? ge[g_index(ch, r, cc)] : narrow<T>(CT(0)); };
        auto gtap = [&](int r, int cc) { return widen<T>(gtap_s(r, cc)); };
        if (!(h >= LH && h < LH + OH && wq >= LW && wq < LW + OW)) {   // outside the window
            oe[(static_cast<int64_t>(h) * W + wq) * C] = narrow<T>(CT(0));
            return;
        }
        const int a0 = fold_w(wq - cxW), a1 = fold_w(wq - cxW + 1);
        const int b0 = fold_gw(wq - LW - cgW) - LW, b1 = fold_gw(wq - LW - cgW + 1) - LW;   // grad_out columns (< 0: padding)
        const int r0 = fold_h(h - cxH), r1 = fold_h(h - cxH + 1);
        const int s0 = fold_gh(h - LH - cgH) - LH, s1 = fold_gh(h - LH - cgH + 1) - LH;   // grad_out rows (< 0: padding)
        CT v[4] = {tap(xe, r0, a0), tap(xe, r1, a0), tap(xe, r0, a1), tap(xe, r1, a1)}, wg[3];
        const CT gval = widen<T>(ge[g_index(ch, h - LH, wq - LW)]);
        weight_grads_nd<2, CT>(v, fr, wg);
        t0 += static_cast<double>(gval * wg[0]);
        t1 += static_cast<double>(gval * wg[1]);
        S r;
        if constexpr (ACTIVE) {
            CT u[4] = {gtap(s0, b0), gtap(s1, b0), gtap(s0, b1), gtap(s1, b1)};
            r = narrow<T>(interp_t<T, 2>(u, fr));
        } else {
            r = gtap_s(s0, b0);
        }
        oe[(static_cast<int64_t>(h) * W + wq) * C] = r;
    };
    // (a) the few elements of channels INSIDE the ring: the reflected corner one step beyond it, the rows / columns whose periodic
    // source wraps -- by the thread that owns them
    bool any_scol = false;
#pragma unroll
    for (int i = 0; i < NI; ++i) any_scol = any_scol || scol[i];
    const bool wrap_rows = periodic && near_c && (h0 <= kR || h1 >= H - kR - 1 || (h0 <= LH + kR && h1 > LH) || (h1 >= LH + OH - kR - 1 && h0 < LH + OH));
    if (near_c && (any_scol || srow || srow_g || wrap_rows)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int wq = w0 + lane_b + PL * i;
            if (wq >= W) continue;
            for (int h = h0; h < h1; ++h) {
                if (!(scol[i] || (srow && h == H - 1) || (srow_g && h == LH + OH - 1) || wraps(h))) continue;
                far_element(c, h, wq, csxH, csxW, csgH, csgW, dw, acc[0], acc[1]);
            }
        }
    }
    // (b) channels whose shift leaves the ring (|row or column shift| > R): ALL threads of the workgroup share the band's elements of
    // such a channel (round 5: its own pixel lanes alone walked the band's rows one after the other -- 56 dependent round trips; one
    // such channel in 256 doubled the kernel's time: N16 C256 224x224 fp32 0.53 -> 0.99 ms)
    {
        __shared__ unsigned long long far_mask_s;
        __syncthreads();
        if (threadIdx.x < 64) {   // (the channel lanes with pixel lane 0 are threads 0 .. CB - 1: inside the first wave)
            const unsigned long long m = __ballot(far_c && static_cast<int>(threadIdx.x) < CB);
            if (threadIdx.x == 0) far_mask_s = m;
        }
        __syncthreads();
        unsigned long long fm_left = far_mask_s;
        double *fred = reinterpret_cast<double *>(ring_all);   // the rings are dead
        while (fm_left) {   // (uniform)
            const int fch = __builtin_ctzll(fm_left);
            fm_left &= fm_left - 1;
            const int cf = c0 + fch;
            int64_t fsh[3] = {0, 0, 0};
            CT ffr[3] = {CT(0), CT(0), CT(0)};
            {
                const int wcol[3] = {-1, 0, 1};
                CT wv[3];
                load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(cf) * 2, wcol, wv);
                prep_shift_backward<CT>(wv[1], ACTIVE, fsh[1], ffr[0]);
                prep_shift_backward<CT>(wv[2], ACTIVE, fsh[2], ffr[1]);
            }
            const int fxH = canon_shift(fsh[1], H, p.pad, p.d_perH), fxW = canon_shift(fsh[2], W, p.pad, p.d_perW);
            const int fgH = canon_shift(ACTIVE ? fsh[1] : -fsh[1], OH, p.pad, p.d_perOH), fgW = canon_shift(ACTIVE ? fsh[2] : -fsh[2], OW, p.pad, p.d_perOW);
            double t0 = 0.0, t1 = 0.0;
            const int nelem = (h1 - h0) * TW;
            for (int q = static_cast<int>(threadIdx.x); q < nelem; q += kThreads) {
                const int h = h0 + q / TW, wq = w0 + q % TW;
                if (wq < W) far_element(cf, h, wq, fxH, fxW, fgH, fgW, ffr, t0, t1);
            }
            __syncthreads();
            fred[threadIdx.x * 2] = t0;
            fred[threadIdx.x * 2 + 1] = t1;
            __syncthreads();
            if (static_cast<int>(threadIdx.x) == fch) {   // the channel's owner (pixel lane 0): the threads' sums in thread order
                double u0 = 0.0, u1 = 0.0;
                for (int k = 0; k < kThreads; ++k) {
                    u0 += fred[k * 2];
                    u1 += fred[k * 2 + 1];
                }
                acc[0] += u0;
                acc[1] += u1;
            }
        }
    }

    // ---- the workgroup's weight-gradient partial sums: pixel lanes of each channel, in lane order -----------------------
    __syncthreads();
    double *red = reinterpret_cast<double *>(ring_all);   // the rings are dead
    red[threadIdx.x * 2] = acc[0];
    red[threadIdx.x * 2 + 1] = acc[1];
    __syncthreads();
    if (lane_b == 0 && live_c) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < PL; ++k) {
            t0 += red[(k * CB + lane_a) * 2];
            t1 += red[(k * CB + lane_a) * 2 + 1];
        }
        double *dst = p.partials + (static_cast<size_t>(pidx) * C + c) * 3;
        dst[0] = t0;
        dst[1] = t1;
        dst[2] = 0.0;
    }
}

// =====================================================================================================
// Active (interpolating) forward for dense channels-last fp32 inputs: out = interp of the four corners around
// (o - floor(w)) (shifts_kernels.h:330-400 with :187-205).  The gather forward's tile (32 columns x 32 channels) with
// a ring of rows h - R .. h + R + 1 and one more staged column for the + 1 corners; output channels-last or
// NCHW-contiguous.  Roofline: HBM, 2 x 4 bytes per element.
// =====================================================================================================
constexpr int kAPW = kTW + 2 * kR + 1;         // staged pixels per row
constexpr int kARing = 8;                      // staged rows h - R .. h + R + 1
constexpr int kARowWords = kAPW * kPitch;
constexpr int kAPieces = kAPW * (kLine / 16);  // 312
constexpr int kANP = (kAPieces + kThreads - 1) / kThreads;

//
// ND3 (round 4, NDHWC): the reference's 3-D blend nests the PLANE blend innermost (interpolation.h:34-40), so it is done at staging
// time: a ring element is u = lerp(x[plane0], x[plane1], d_plane) of its channel's two source planes, in fp32 whatever the tensor's
// type (32 channels per workgroup), and everything behind the staging is the 2-D kernel on u with the row / column fractions --
// the same bits as interp_nd<3>.  One output plane per workgroup, the plane the fastest block index (cl_tiled_forward<.., ND3>).
template <typename T, bool OUT_CL, bool ND3>
__global__ __launch_bounds__(kThreads) void cl_tiled_active_forward(const ClTiledParams p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    using S = typename T::S;
    using CT = typename T::C;
    static_assert(sizeof(S) == 4 || sizeof(S) == 2, "fp32, fp16, bf16");
    static_assert(!ND3 || sizeof(CT) == 4, "the ring of the 3-D form holds fp32 plane blends");
    constexpr int ES = sizeof(S), RES = ND3 ? 4 : ES;   // element size in memory / in the ring
    constexpr int CB = kLine / RES;                     // channels per workgroup: 32 (fp32, ND3) or 64 (16-bit types, 2-D)
    // OUT_CL: thread = (channel lane % CB, pixel lane), columns pl + PL i;  else thread = (column lane % 32, channel lane
    // 0..7), channels cl + 8 i.  Either way NI outputs per row.
    constexpr int LA = OUT_CL ? CB : kTW, PL = kThreads / LA, NI = OUT_CL ? kTW / PL : CB / PL;
    // four zero words (what a padding corner reads: cl_tiled_backward), the ring, dump words
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[4 + kARing * kARowWords + 4];
    uint32_t *const ring = ring_all + 4;
    constexpr int kNeg = -(1 << 24);
    if (threadIdx.x < 4) ring_all[threadIdx.x] = 0u;
    __shared__ int tab_sh[CB], tab_sw[CB];     // signed row shift (or out of the ring: INT_MIN), canonical column shift
    __shared__ int tab_shc[CB];                // canonical row shift (the element-by-element pass)
    __shared__ float tab_dh[CB], tab_dw[CB];   // interpolation fractions
    __shared__ int tab_pz0[ND3 ? CB : 1], tab_pz1[ND3 ? CB : 1];   // ND3: the channel's two source planes (-1: padding)
    __shared__ float tab_dp[ND3 ? CB : 1];                          // ... and its plane fraction
    constexpr int kDump = kARing * kARowWords;
    constexpr int kFarShift = -0x7fffffff - 1;

    // XCD-contiguous ids: workgroups that share an XCD (blockIdx % 8) and its L2 own neighbouring tiles (shared halo)
    unsigned b = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    int dz = 0;   // ND3: the output plane, the fastest index (neighbouring planes share their pixel lines in L2)
    if constexpr (ND3) {
        const unsigned q = fdiv(b, p.d_OD);
        dz = static_cast<int>(b - q * static_cast<unsigned>(p.OD));
        b = q;
    }
    const int wt = static_cast<int>(b - fdiv(b, p.d_wtiles) * p.wtiles);
    b = fdiv(b, p.d_wtiles);
    const int cb = static_cast<int>(b - fdiv(b, p.d_cblocks) * p.cblocks);
    b = fdiv(b, p.d_cblocks);
    const int band = static_cast<int>(b - fdiv(b, p.d_bands) * p.bands);
    const int n = static_cast<int>(fdiv(b, p.d_bands));
    const int w0 = wt * kTW, c0 = cb * CB;
    // the window (round 4, as in cl_tiled_forward): output row h / column w read the source around (h + LH, w + LW)
    const int H = p.H, W = p.W, C = p.C, OH = p.OH, OW = p.OW, LH = p.LH, LW = p.LW;
    const int h0 = band * p.band_rows, h1 = min(OH, h0 + p.band_rows);
    const int DZ = ND3 ? p.D : 1, OD = ND3 ? p.OD : 1;
    const char *xn = p.x + static_cast<int64_t>(n) * DZ * H * W * C * ES;
    char *on = p.out + static_cast<int64_t>(n) * OD * OH * OW * C * ES;
    const uint32_t plane_bytes = static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * static_cast<uint32_t>(C) * ES;
    const uint32_t img_bytes = static_cast<uint32_t>(DZ) * plane_bytes;
    const uint32_t out_bytes = static_cast<uint32_t>(OD) * static_cast<uint32_t>(OH) * static_cast<uint32_t>(OW) * static_cast<uint32_t>(C) * ES;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xn), 0, img_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(on, 0, out_bytes, kBufferFlags);

    // ---- the channels' shifts (ND3: first -- the staging addresses depend on the depth shift) ----------------------------------
    const int perH = map_period(H, p.pad), perW = map_period(W, p.pad);
    auto channel_table = [&]() {
        if (threadIdx.x < CB) {
            const int c = min(c0 + static_cast<int>(threadIdx.x), C - 1);
            CT wv[3], dh, dwf;
            int64_t sh2[2];
            if constexpr (ND3) {
                const int wcol[3] = {0, 1, 2};
                load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 3, wcol, wv);
                int64_t shd;
                CT dp;
                prep_shift_forward<CT>(wv[0], true, shd, dp);
                const int sd = canon_shift(shd, DZ, p.pad, p.d_perD);
                tab_pz0[threadIdx.x] = DZ == 1 ? 0 : fold_index(dz + p.LD - sd, DZ, p.pad);   // size-1 dims ignore the shift
                tab_pz1[threadIdx.x] = DZ == 1 ? 0 : fold_index(dz + p.LD - sd + 1, DZ, p.pad);
                tab_dp[threadIdx.x] = dp;
            } else {
                const int wcol[3] = {-1, 0, 1};
                load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(c) * 2, wcol, wv);
            }
            prep_shift_forward<CT>(wv[1], true, sh2[0], dh);
            prep_shift_forward<CT>(wv[2], true, sh2[1], dwf);
            const int sh = canon_shift(sh2[0], H, p.pad, p.d_perH), sw = canon_shift(sh2[1], W, p.pad, p.d_perW);
            const int sh_s = (perH && 2 * sh > perH) ? sh - perH : sh, sw_s = (perW && 2 * sw > perW) ? sw - perW : sw;
            const bool in_ring = sh_s >= -kR && sh_s <= kR && sw_s >= -kR && sw_s <= kR;
            tab_sh[threadIdx.x] = in_ring ? sh_s : kFarShift;
            tab_shc[threadIdx.x] = sh;
            tab_sw[threadIdx.x] = sw;
            tab_dh[threadIdx.x] = dh;
            tab_dw[threadIdx.x] = dwf;
        }
        __syncthreads();
    };
    if constexpr (ND3) channel_table();

    // ---- staging (as in cl_tiled_forward; ND3: element px * 32 + channel of the staged row from the channel's two planes) -----
    constexpr int kNS = ND3 ? (kAPW * CB + kThreads - 1) / kThreads : kANP;
    struct Pair { uint32_t a, b; };
    using SV = std::conditional_t<ND3, Pair, u4>;
    uint32_t poff[kNS], poff1[ND3 ? kNS : 1];
    int pdst[kNS];
    float my_dp = 0.f;   // ND3: the plane fraction of the thread's staging channel (element q: channel q % 32 = tid % 32 for every k)
    if constexpr (ND3) my_dp = tab_dp[threadIdx.x % CB];
#pragma unroll
    for (int k = 0; k < kNS; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        if constexpr (ND3) {
            const int px = q / CB, ch = q - px * CB, gx = w0 + LW - kR + px;
            const bool elem = px < kAPW && gx >= 0 && gx < W && c0 + ch < C;
            const int pz0 = tab_pz0[ch], pz1 = tab_pz1[ch];
            const uint32_t inplane = (static_cast<uint32_t>(gx) * C + c0 + ch) * ES;
            poff[k] = (elem && pz0 >= 0) ? static_cast<uint32_t>(pz0) * plane_bytes + inplane : kOutOfRange;
            poff1[k] = (elem && pz1 >= 0) ? static_cast<uint32_t>(pz1) * plane_bytes + inplane : kOutOfRange;
            pdst[k] = px < kAPW ? px * (kPitch * 4) + ch * 4 : -1;   // (bytes)
        } else {
            const int px = q >> 3, cbyte = c0 * ES + (q & 7) * 16, gx = w0 + LW - kR + px;
            const bool piece = q < kAPieces;
            poff[k] = (piece && gx >= 0 && gx < W && cbyte < C * ES) ? static_cast<uint32_t>(gx) * C * ES + cbyte : kOutOfRange;
            pdst[k] = piece ? px * kPitch + (q & 7) * 4 : -1;
        }
    }
    const uint32_t row_bytes = static_cast<uint32_t>(W) * C * ES;
    constexpr int kDepth = CLT_DEPTH;
    SV pvs[kDepth][kNS];
    auto load_elem = [&](uint32_t off, uint32_t so) {
        if constexpr (ES == 4) return static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b32(xres, off, so, 0));
        else return static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b16(xres, off, so, 0));
    };
    auto load_row = [&](int y, int ylast, SV (&pv)[kNS]) {
        const int unwanted = (y >> 31) | ((ylast - y) >> 31);   // (sign arithmetic: cl_tiled_backward)
        const uint32_t so = static_cast<uint32_t>(y & ~unwanted) * row_bytes, dead = static_cast<uint32_t>(unwanted) & kOutOfRange;
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            if constexpr (ND3) {
                pv[k].a = load_elem(poff[k] | dead, so);
                pv[k].b = load_elem(poff1[k] | dead, so);
            } else {
                pv[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, poff[k] | dead, so, 0);
            }
        }
    };
    auto as_ct = [](uint32_t bits) {   // a loaded element (raw bits in the low half for 16-bit types) in the compute type
        if constexpr (ES == 4) return __builtin_bit_cast(float, bits);
        else return widen<T>(__builtin_bit_cast(S, static_cast<uint16_t>(bits)));
    };
    auto store_row = [&](int y, const SV (&pv)[kNS]) {
        const int slot = y & (kARing - 1);
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            if constexpr (ND3) {
                // the plane blend, exactly the innermost blend of interp_nd<3> / interp_nd_fused<3>
                const CT v[2] = {as_ct(pv[k].a), as_ct(pv[k].b)}, d1[1] = {my_dp};
                const CT u = interp_t<T, 1>(v, d1);
                char *d = reinterpret_cast<char *>(ring) + (pdst[k] >= 0 ? slot * (kARowWords * 4) + pdst[k] : kDump * 4);
                *reinterpret_cast<float *>(d) = u;
            } else {
                uint32_t *d = ring + (pdst[k] >= 0 ? slot * kARowWords + pdst[k] : kDump);
                d[0] = pv[k].x;
                d[1] = pv[k].y;
                d[2] = pv[k].z;
                d[3] = pv[k].w;
            }
        }
    };
    const int ylast = min(H - 1, h1 + LH + kR);
    SV pre[2 * kR + 1][kNS];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_row(h0 + LH + kR + 1 + d, ylast, pvs[d]);
#pragma unroll
    for (int r = 0; r <= 2 * kR; ++r) load_row(h0 + LH - kR + r, ylast, pre[r]);

    if constexpr (!ND3) channel_table();   // (2-D: while the rows are in flight)

    // ---- thread -> outputs ---------------------------------------------------------------------------------------------
    const int lane_a = static_cast<int>(threadIdx.x) % LA, lane_b = static_cast<int>(threadIdx.x) / LA;
    constexpr int NCH = OUT_CL ? 1 : NI;
    auto fold_w = [&](int idx) { return W == 1 ? 0 : fold_index(idx, W, p.pad); };
    auto fold_h = [&](int idx) { return H == 1 ? 0 : fold_index(idx, H, p.pad); };
    int shs[NCH];              // signed row shift of the thread's channel(s)
    CT dws[NCH][2];
    bool srow[NCH];
    int xc0[NI], xc1[NI];      // LDS byte offsets (within a ring row) of the corners' columns; -1: padding
    uint32_t ooff[NI];
    uint32_t live = 0, rest = 0;   // bit i: served from the ring / left to the element-by-element pass (all rows)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int ch = OUT_CL ? lane_a : lane_b + PL * i, col = OUT_CL ? lane_b + PL * i : lane_a;
        const int c = c0 + ch, k = OUT_CL ? 0 : i;
        const bool inside = c < C && w0 + col < OW;
        const int s = tab_sh[ch], sw = tab_sw[ch];
        shs[k] = s == kFarShift ? 0 : s;
        dws[k][0] = tab_dh[ch];
        dws[k][1] = tab_dw[ch];
        // the one source row the ring cannot hold: reflect padding, last output row, shift -R (see cl_tiled_backward)
        srow[k] = c < C && s != kFarShift && p.pad == 3 && H > 1 && s == -kR && h1 + LH == H;
        const int wq = w0 + LW + min(col, OW - 1 - w0);   // the source column under a zero shift
        const int a0 = fold_w(wq - sw), a1 = fold_w(wq - sw + 1);
        auto outside = [&](int sx) { return sx >= 0 && (sx < w0 + LW - kR || sx > w0 + LW + kTW + kR); };
        const bool mine = inside && s != kFarShift && !outside(a0) && !outside(a1);
        live |= (mine ? 1u : 0u) << i;
        rest |= ((inside && !mine) ? 1u : 0u) << i;
        auto lds_col = [&](int sx) { return (mine && sx >= 0) ? (sx - (w0 + LW - kR)) * (kPitch * 4) + ch * RES : kNeg; };
        xc0[i] = lds_col(a0);
        xc1[i] = lds_col(a1);
        const uint32_t o = OUT_CL ? (static_cast<uint32_t>((dz * OH + h0) * OW + w0 + col) * C + c) * ES
                                  : (static_cast<uint32_t>((c * OD + dz) * OH + h0) * OW + w0 + col) * ES;
        ooff[i] = mine ? o : kOutOfRange;
    }
    const uint32_t ostep = static_cast<uint32_t>(OUT_CL ? OW * C : OW) * ES;

#pragma unroll
    for (int r = 0; r <= 2 * kR; ++r) {
        const int y = h0 + LH - kR + r;
        if (y >= 0 && y < H) store_row(y, pre[r]);
    }
    const char *ringz = reinterpret_cast<const char *>(ring_all);   // byte 0: the zero words; the ring starts at byte 16
    auto lds_f = [&](int row, int colo) {   // padding (a negative offset) reads a zero word: no lane masks
        if constexpr (ND3) return *reinterpret_cast<const float *>(ringz + max(row + colo, 0));
        else return widen<T>(*reinterpret_cast<const S *>(ringz + max(row + colo, 0)));
    };
    auto row_off = [&](int sy) { return (16 + (sy & (kARing - 1)) * (kARowWords * 4)) | ((sy >> 31) & kNeg); };
    const int fm = (p.pad == 3 || p.pad == 4) ? -1 : 0;   // one fold of the signed shift, in bit arithmetic: see cl_tiled_backward
    const int fLo = p.pad == 4 ? -1 : 0, fHi = p.pad == 1 ? H - 1 : (p.pad == 3 ? 2 * H - 2 : 2 * H - 1);
    const int zneg = p.pad == 0 ? kNeg : 0;
    auto fold1 = [&](int idx) {
        const int below = idx >> 31, above = (H - 1 - idx) >> 31, t = idx & fm;
        const int r = (idx & ~(below | above)) | ((fLo - t) & below) | ((fHi - t) & above) | ((below | above) & zneg);
        return H == 1 ? 0 : r;
    };
    // periodic padding: a corner row that wraps to the far side of the image is not in the ring; such rows (at most R + 1 at
    // the top or the bottom of the image) are left to the element-by-element pass, like the reflected corner above
    const bool periodic = p.pad == 2 && H > 1;
    auto wraps = [&](int hs, int sgn) { return periodic && (hs - sgn < 0 || hs + 1 - sgn >= H); };   // hs: the source row h + LH
    int xrow1[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) xrow1[k] = fold1(h0 + LH - shs[k]);
    auto step = [&](int h, SV (&pv)[kNS]) {
        const int hs = h + LH;
        __syncthreads();
        if (hs + kR + 1 < H) store_row(hs + kR + 1, pv);
        __syncthreads();
        load_row(hs + kDepth + kR + 1, ylast, pv);
        const uint32_t so = static_cast<uint32_t>(h - h0) * ostep;
        int xr0[NCH], xr1[NCH];
        int skip[NCH];   // -1: the row is left to the element pass (launch-uniform guards: one padding mode each)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            xr0[k] = row_off(xrow1[k]);
            xrow1[k] = fold1(hs + 1 - shs[k]);
            xr1[k] = row_off(xrow1[k]);
            skip[k] = 0;
            if (p.pad == 3) skip[k] = (srow[k] && hs == H - 1) ? -1 : 0;
            else if (p.pad == 2) skip[k] = wraps(hs, shs[k]) ? -1 : 0;
        }
        S res[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = OUT_CL ? 0 : i;
            CT v[4];
            v[0] = lds_f(xr0[k], xc0[i]);
            v[1] = lds_f(xr1[k], xc0[i]);
            v[2] = lds_f(xr0[k], xc1[i]);
            v[3] = lds_f(xr1[k], xc1[i]);
            res[i] = narrow<T>(interp_t<T, 2>(v, dws[k]));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t off = ooff[i] | (static_cast<uint32_t>(skip[OUT_CL ? 0 : i]) & kOutOfRange);
            if constexpr (ES == 4) {
                uint32_t bits;
                __builtin_memcpy(&bits, &res[i], 4);
                __builtin_amdgcn_raw_buffer_store_b32(bits, ores, off, so, 0);
            } else {
                uint16_t bits;
                __builtin_memcpy(&bits, &res[i], 2);
                __builtin_amdgcn_raw_buffer_store_b16(bits, ores, off, so, 0);
            }
        }
    };
    int hb = h0;
    for (; hb + kDepth <= h1; hb += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) step(hb + d, pvs[d]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)
        if (hb + d < h1) step(hb + d, pvs[d]);

    // ---- what the ring could not serve, element by element from memory (rare) --------------------------------------------
    bool any_srow = false;
#pragma unroll
    for (int k = 0; k < NCH; ++k) any_srow = any_srow || srow[k];
    const bool wrap_rows = periodic && (h0 + LH <= kR || h1 + LH >= H - kR - 1);
    if (rest || any_srow || (wrap_rows && live)) {
        const S *xe = reinterpret_cast<const S *>(xn);
        int ez0 = 0, ez1 = 0;   // ND3: the two source planes and the plane fraction of the element at hand
        CT edp = CT(0);
        const int64_t plane_elems = static_cast<int64_t>(H) * W * C;
        auto tap = [&](const S *base, int r, int cc) {
            if (r < 0 || cc < 0) return CT(0);
            const int64_t at = (static_cast<int64_t>(r) * W + cc) * C;
            if constexpr (ND3) {   // the plane blend first, like the staging
                const CT v[2] = {ez0 >= 0 ? widen<T>(base[ez0 * plane_elems + at]) : CT(0), ez1 >= 0 ? widen<T>(base[ez1 * plane_elems + at]) : CT(0)};
                const CT d1[1] = {edp};
                return interp_t<T, 1>(v, d1);
            } else {
                return widen<T>(base[at]);
            }
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = OUT_CL ? 0 : i;
            const bool all_rows = (rest >> i) & 1u, mine = (live >> i) & 1u;
            if (!all_rows && !(mine && (srow[k] || wrap_rows))) continue;
            const int ch = OUT_CL ? lane_a : lane_b + PL * i, col = OUT_CL ? lane_b + PL * i : lane_a;
            const int shc = tab_shc[ch], sw = tab_sw[ch], wo = w0 + col, wq = wo + LW;
            const int a0 = fold_w(wq - sw), a1 = fold_w(wq - sw + 1);
            if constexpr (ND3) {
                ez0 = tab_pz0[ch];
                ez1 = tab_pz1[ch];
                edp = tab_dp[ch];
            }
            S *o = reinterpret_cast<S *>(on) + (OUT_CL ? ((static_cast<int64_t>(dz) * OH + h0) * OW + wo) * C + c0 + ch
                                                               : ((static_cast<int64_t>(c0 + ch) * OD + dz) * OH + h0) * OW + wo);
            for (int h = h0; h < h1; ++h) {
                const int hs = h + LH;
                if (!all_rows && !((srow[k] && hs == H - 1) || wraps(hs, shs[k]))) continue;
                const int r0 = fold_h(hs - shc), r1 = fold_h(hs - shc + 1);
                CT v[4] = {tap(xe + c0 + ch, r0, a0), tap(xe + c0 + ch, r1, a0), tap(xe + c0 + ch, r0, a1), tap(xe + c0 + ch, r1, a1)};
                o[static_cast<int64_t>(h - h0) * (OUT_CL ? OW * C : OW)] = narrow<T>(interp_t<T, 2>(v, dws[k]));
            }
        }
    }
}

// [0] enabled, [1] rows per band (0 = automatic), [2] XCD-contiguous block ids (N16 C256 224x224 fp32: forward to NCHW
// 0.320 -> 0.292 ms, backward 0.58 -> 0.525 ms: neighbouring tiles share their halo in one L2)
thread_local int g_cl_tiled_tune[3] = {1, 0, 1};

bool dense_channels_last_2d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    // normalised strides N, C, d0, d1, inner with d0 of size 1
    return st[1] == 1 && st[4] == g.C && (sz[1] == 1 || st[3] == g.C * sz[2]) && (g.N == 1 || st[0] == g.C * sz[1] * sz[2]);
}
bool contiguous_2d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[4] == 1 && (sz[1] == 1 || st[3] == sz[2]) && st[1] == sz[1] * sz[2] && (g.N == 1 || st[0] == g.C * sz[1] * sz[2]);
}
// 3-D: dense NDHWC (channels_last_3d) / NCDHW-contiguous
bool dense_channels_last_3d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[1] == 1 && st[4] == g.C && (sz[1] == 1 || st[3] == g.C * sz[2]) && (sz[0] == 1 || st[2] == g.C * sz[1] * sz[2]) &&
           (g.N == 1 || st[0] == g.C * sz[0] * sz[1] * sz[2]);
}
bool contiguous_3d(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[4] == 1 && (sz[1] == 1 || st[3] == sz[2]) && (sz[0] == 1 || st[2] == sz[1] * sz[2]) && st[1] == sz[0] * sz[1] * sz[2] &&
           (g.N == 1 || st[0] == g.C * sz[0] * sz[1] * sz[2]);
}

}  // namespace

void cl_tiled_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 3) g_cl_tiled_tune[knob] = value;
}

// 2-D, 1- / 2- / 4-byte elements, pure gather (sparse shift / quantized), no crop, dense channels-last
// input whose pixel lines (C elements) are whole 16-byte pieces, 16-byte aligned; output dense channels-last or
// NCHW-contiguous with rows of whole dwords
bool cl_tiled_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    const int es = dtype_size(dtype);
    if (!g_cl_tiled_tune[0] || (g.nd != 2 && g.nd != 3) || es > 4) return false;
    if (g.active && dtype <= SHIFTND_BF16) {  // interpolating: cl_tiled_active_forward (fp32, fp16, bf16), rows folded once
        if (dtype == SHIFTND_F64 || (g.S[1] != 1 && g.S[1] < 5)) return false;
    }
    if (g.S[1] != 1 && g.S[1] <= kR) return false;   // the gather kernel folds its source rows once too (round 4)
    for (int d = 0; d < 3; ++d)   // the window (a crop of the output, round 4): both forward kernels
        if ((g.L[d] != 0 || g.O[d] != g.S[d]) && ((d == 0 && g.nd != 3) || g.O[d] < 1)) return false;
    if ((g.C * es) % 16 != 0 || g.S[1] >= (1 << 20) || g.S[2] >= (1 << 20) || g.N >= (1LL << 24) || g.C >= (1 << 24)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 4 != 0) return false;
    if (g.nd == 3) {   // NDHWC (round 4): the gather kernel of 2- and 4-byte elements, one output plane per workgroup
        if (es < 2 || g.S[0] >= (1 << 20) || g.N * g.O[0] >= (1LL << 24)) return false;
        if (g.C * g.S[0] * g.S[1] * g.S[2] * es >= (1LL << 31) || g.C * g.O[0] * g.O[1] * g.O[2] * es >= (1LL << 31)) return false;
        if (!dense_channels_last_3d(g.xs, g, g.S)) return false;
        if (dense_channels_last_3d(g.os, g, g.O)) return true;
        return contiguous_3d(g.os, g, g.O) && (g.O[2] * es) % 4 == 0;
    }
    if (g.C * g.S[1] * g.S[2] * es >= (1LL << 31)) return false;  // one image per buffer resource, offsets below 2^31
    if (!dense_channels_last_2d(g.xs, g, g.S)) return false;
    if (dense_channels_last_2d(g.os, g, g.O)) return true;
    return contiguous_2d(g.os, g, g.O) && (g.O[2] * es) % 4 == 0;
}

int cl_tiled_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                     void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    ClTiledParams p{};
    p.x = static_cast<const char *>(x);
    p.out = static_cast<char *>(out);
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = static_cast<uint32_t>(es == 4 ? fill_bits : (fill_bits & ((1ull << (8 * es)) - 1)));
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.H = static_cast<int>(g.S[1]);
    p.W = static_cast<int>(g.S[2]);
    p.OH = static_cast<int>(g.O[1]);
    p.OW = static_cast<int>(g.O[2]);
    p.LH = static_cast<int>(g.L[1]);
    p.LW = static_cast<int>(g.L[2]);
    p.D = static_cast<int>(g.S[0]);
    p.OD = static_cast<int>(g.O[0]);
    p.LD = static_cast<int>(g.L[0]);
    p.pad = g.pad;
    const bool nd3 = g.nd == 3;
    p.out_cl = (nd3 ? dense_channels_last_3d(g.os, g, g.O) : dense_channels_last_2d(g.os, g, g.O)) ? 1 : 0;
    p.wtiles = (p.OW + kTW - 1) / kTW;
    const bool active3 = nd3 && g.active && dtype <= SHIFTND_BF16;   // (its ring holds fp32 plane blends: 32 channels per workgroup)
    const int cb = active3 ? kLine / 4 : kLine / es;
    p.cblocks = (p.C + cb - 1) / cb;
    // bands along H: enough workgroups (~7 per workgroup slot of the chip: 28-row bands measured best on N16 C256
    // 224x224), at least 8 R rows per band (the ring warm-up is 2 R rows)
    const int64_t base = static_cast<int64_t>(p.N) * p.OD * p.wtiles * p.cblocks;   // (2-D: OD == 1)
    int64_t bands = g_cl_tiled_tune[1] > 0 ? (p.OH + g_cl_tiled_tune[1] - 1) / g_cl_tiled_tune[1] : (7168 + base - 1) / base;
    const int64_t max_bands = p.OH / (8 * kR) > 0 ? p.OH / (8 * kR) : 1;
    if (g_cl_tiled_tune[1] <= 0 && bands > max_bands) bands = max_bands;
    if (bands < 1) bands = 1;
    p.band_rows = static_cast<int>((p.OH + bands - 1) / bands);
    p.bands = (p.OH + p.band_rows - 1) / p.band_rows;
    const int64_t grid = base * p.bands;
    if (grid >= (1LL << 31)) return SHIFTND_ERR_TOO_LARGE;
    p.xcd_blocks = (g_cl_tiled_tune[2] && grid % 8 == 0) ? static_cast<unsigned>(grid / 8) : 0;
    p.d_wtiles = make_fastdiv(static_cast<uint32_t>(p.wtiles));
    p.d_cblocks = make_fastdiv(static_cast<uint32_t>(p.cblocks));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_perH = make_fastdiv(static_cast<uint32_t>(map_period(p.H, p.pad)));
    p.d_perW = make_fastdiv(static_cast<uint32_t>(map_period(p.W, p.pad)));
    p.d_perD = make_fastdiv(static_cast<uint32_t>(map_period(p.D, p.pad)));
    p.d_OD = make_fastdiv(static_cast<uint32_t>(p.OD));
    if (active3) {
        note_kernel("cl_tiled_active_forward_3d");
#define SHIFTND_CLT_ACTIVE3(TT) \
    if (p.out_cl) hipLaunchKernelGGL((cl_tiled_active_forward<TT, true, true>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p); \
    else hipLaunchKernelGGL((cl_tiled_active_forward<TT, false, true>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
        if (dtype == SHIFTND_F32) { SHIFTND_CLT_ACTIVE3(f32_t) }
        else if (dtype == SHIFTND_F16) { SHIFTND_CLT_ACTIVE3(f16_t) }
        else { SHIFTND_CLT_ACTIVE3(bf16_t) }
#undef SHIFTND_CLT_ACTIVE3
        return SHIFTND_OK;
    }
    if (nd3) {
        note_kernel("cl_tiled_forward_3d");
#define SHIFTND_CLT_LAUNCH3(ESV) \
    if (p.out_cl) hipLaunchKernelGGL((cl_tiled_forward<ESV, true, true>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p); \
    else hipLaunchKernelGGL((cl_tiled_forward<ESV, false, true>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
        if (es == 4) { SHIFTND_CLT_LAUNCH3(4) }
        else { SHIFTND_CLT_LAUNCH3(2) }
#undef SHIFTND_CLT_LAUNCH3
        return SHIFTND_OK;
    }
    if (g.active && dtype <= SHIFTND_BF16) {
        note_kernel("cl_tiled_active_forward");
#define SHIFTND_CLT_ACTIVE(TT) \
    if (p.out_cl) hipLaunchKernelGGL((cl_tiled_active_forward<TT, true, false>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p); \
    else hipLaunchKernelGGL((cl_tiled_active_forward<TT, false, false>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
        if (dtype == SHIFTND_F32) { SHIFTND_CLT_ACTIVE(f32_t) }
        else if (dtype == SHIFTND_F16) { SHIFTND_CLT_ACTIVE(f16_t) }
        else { SHIFTND_CLT_ACTIVE(bf16_t) }
#undef SHIFTND_CLT_ACTIVE
        return SHIFTND_OK;
    }
    note_kernel("cl_tiled_forward");
#define SHIFTND_CLT_LAUNCH(ESV) \
    if (p.out_cl) hipLaunchKernelGGL((cl_tiled_forward<ESV, true, false>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p); \
    else hipLaunchKernelGGL((cl_tiled_forward<ESV, false, false>), dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, st, p);
    if (es == 4) { SHIFTND_CLT_LAUNCH(4) }
    else if (es == 2) { SHIFTND_CLT_LAUNCH(2) }
    else { SHIFTND_CLT_LAUNCH(1) }
#undef SHIFTND_CLT_LAUNCH
    return SHIFTND_OK;
}

// ---- backward ------------------------------------------------------------------------------------------------------
namespace {
struct ClTiledBwdPlan {
    int wtiles, cblocks, bands, band_rows, tw;
    int64_t groups;
};
// tw: grad_x columns per workgroup (16; 32: the wide strip, round 6 -- a run's choice, the workspace is planned with 16: more groups)
ClTiledBwdPlan cl_tiled_backward_plan(const Geometry &g, int es = 4, int tw = kBTW) {
    ClTiledBwdPlan pl;
    const int H = static_cast<int>(g.S[1]), W = static_cast<int>(g.S[2]);
    pl.tw = tw;
    pl.wtiles = (W + tw - 1) / tw;
    const int cbw = kLine / es;
    pl.cblocks = static_cast<int>((g.C + cbw - 1) / cbw);
    // bands along H: ~7 workgroups per workgroup slot, at least 8 R rows per band, and (when the batch allows) at most
    // 4096 partial-sum groups per channel
    const int64_t base = g.N * pl.wtiles;
    int64_t bands = g_cl_tiled_tune[1] > 0 ? (H + g_cl_tiled_tune[1] - 1) / g_cl_tiled_tune[1] : (7168 + base * pl.cblocks - 1) / (base * pl.cblocks);
    const int64_t max_bands = H / (8 * kR) > 0 ? H / (8 * kR) : 1;
    if (g_cl_tiled_tune[1] <= 0) {
        if (bands > max_bands) bands = max_bands;
        while (bands > 1 && base * bands > 4096) --bands;
    }
    if (bands < 1) bands = 1;
    pl.band_rows = static_cast<int>((H + bands - 1) / bands);
    pl.bands = (H + pl.band_rows - 1) / pl.band_rows;
    pl.groups = base * pl.bands;
    return pl;
}
}  // namespace

// 2-D fp32 / fp16 / bf16, no crop; saved input and grad_x dense channels-last, incoming gradient too or NCHW-contiguous, pixel
// lines of whole 16-byte pieces
bool cl_tiled_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (!g_cl_tiled_tune[0] || g.nd != 2 || (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16)) return false;
    const int es = dtype_size(dtype);
    for (int d = 0; d < 3; ++d)   // the window (round 4): grad_out has its sizes
        if ((g.L[d] != 0 || g.O[d] != g.S[d]) && (d == 0 || g.O[d] < 1)) return false;
    if ((g.C * es) % 16 != 0 || g.S[1] >= (1 << 20) || g.S[2] >= (1 << 20) || g.N >= (1LL << 24) || g.C >= (1 << 24)) return false;
    if ((g.S[1] != 1 && g.S[1] < 5) || (g.O[1] != 1 && g.O[1] < 5)) return false;  // the kernel folds source rows once
    if (g.C * g.S[1] * g.S[2] * es >= (1LL << 31)) return false;
    if (!dense_channels_last_2d(g.xs, g, g.S) || !dense_channels_last_2d(g.gs, g, g.S)) return false;
    // the incoming gradient: channels-last like the others, or NCHW-contiguous (the mixed form: GO_NCHW)
    const bool go_cl = dense_channels_last_2d(g.os, g, g.O);
    if (!go_cl && !contiguous_2d(g.os, g, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(go) % (go_cl ? 16 : es) != 0 || reinterpret_cast<uintptr_t>(gx) % es != 0) return false;
    const ClTiledBwdPlan pl = cl_tiled_backward_plan(g, es);
    return pl.groups * pl.cblocks < (1LL << 31);
}

size_t cl_tiled_backward_workspace(const Geometry &g) {   // (geometry only: the larger of the 4- and 2-byte plans)
    if (g.nd != 2 || g.C < 1 || g.N * g.S[1] * g.S[2] < 1) return 0;
    const ClTiledBwdPlan p4 = cl_tiled_backward_plan(g, 4), p2 = cl_tiled_backward_plan(g, 2);
    return static_cast<size_t>(p4.groups > p2.groups ? p4.groups : p2.groups) * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

namespace {
template <typename T>
void launch_cl_tiled_backward(const ClTiledBwdParams &p, const ClTiledBwdPlan &pl, bool active, void *gw, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(pl.groups * pl.cblocks)), block(kThreads);
    if (p.go_nchw) {
        if (active) hipLaunchKernelGGL((cl_tiled_backward<T, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((cl_tiled_backward<T, false, true>), grid, block, 0, st, p);
    } else {
        if (active) hipLaunchKernelGGL((cl_tiled_backward<T, true, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((cl_tiled_backward<T, false, false>), grid, block, 0, st, p);
    }
    reduce_weight_grads_of<T>(p.partials, static_cast<int>(pl.groups), p.C, 2, gw, st);
}
}  // namespace

int cl_tiled_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                      void *workspace, hipStream_t st) {
    // (the wide strip, TW = 32 -- built and measured in round 6, same box, fp32 NHWC tensors, bit-identical grad_x over 360 shape /
    //  padding / shift / window combinations: N16 C256 224x224 sparse 0.5115 -> 0.5216 ms, interpolating 0.5628 -> 0.5977; N32 C256
    //  112x112 0.2875 -> 0.3043 / 0.3153 -> 0.3496; N64 C512 56x56 0.3018 -> 0.2983 / 0.3398 -> 0.3435.  The halo shrinks from 1.44 x to
    //  1.22 x of the loads, but 80 KB of LDS and 211 - 227 VGPRs leave two workgroups per CU instead of three: no gain, not instantiated;
    //  the kernel keeps its TW parameter)
    const ClTiledBwdPlan pl = cl_tiled_backward_plan(g, dtype_size(dtype), kBTW);
    ClTiledBwdParams p{};
    p.x = static_cast<const char *>(x);
    p.go = static_cast<const char *>(go);
    p.gx = static_cast<char *>(gx);
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.H = static_cast<int>(g.S[1]);
    p.W = static_cast<int>(g.S[2]);
    p.OH = static_cast<int>(g.O[1]);
    p.OW = static_cast<int>(g.O[2]);
    p.LH = static_cast<int>(g.L[1]);
    p.LW = static_cast<int>(g.L[2]);
    p.pad = g.pad;
    p.go_nchw = dense_channels_last_2d(g.os, g, g.O) ? 0 : 1;   // (a tensor that is both -- C == 1 -- reads the same either way)
    p.go_pieces = (p.go_nchw && (g.O[2] * dtype_size(dtype)) % 16 == 0 && reinterpret_cast<uintptr_t>(go) % 16 == 0) ? 1 : 0;
    p.wtiles = pl.wtiles;
    p.cblocks = pl.cblocks;
    p.bands = pl.bands;
    p.band_rows = pl.band_rows;
    p.d_wtiles = make_fastdiv(static_cast<uint32_t>(p.wtiles));
    p.d_cblocks = make_fastdiv(static_cast<uint32_t>(p.cblocks));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_perH = make_fastdiv(static_cast<uint32_t>(map_period(p.H, p.pad)));
    p.d_perW = make_fastdiv(static_cast<uint32_t>(map_period(p.W, p.pad)));
    p.d_perOH = make_fastdiv(static_cast<uint32_t>(map_period(p.OH, p.pad)));
    p.d_perOW = make_fastdiv(static_cast<uint32_t>(map_period(p.OW, p.pad)));
    {
        const int64_t grid = pl.groups * pl.cblocks;
        p.xcd_blocks = (g_cl_tiled_tune[2] && grid % 8 == 0) ? static_cast<unsigned>(grid / 8) : 0;
    }
    note_kernel(p.go_nchw ? "cl_tiled_backward_nchw_grad" : "cl_tiled_backward");
    switch (dtype) {
    case SHIFTND_F32: launch_cl_tiled_backward<f32_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_F16: launch_cl_tiled_backward<f16_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_BF16: launch_cl_tiled_backward<bf16_t>(p, pl, g.active != 0, gw, st); break;
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
    return SHIFTND_OK;
}

}  // namespace shiftnd
