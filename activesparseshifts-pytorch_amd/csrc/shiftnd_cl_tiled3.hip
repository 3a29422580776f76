// shiftnd_cl_tiled3.hip -- the backward pass of dense NDHWC (channels_last_3d) tensors, LDS-tiled, gfx950 (MI355X); round 5.
// SURVEY section 8f N3; DESIGN section 3.11.  Reference behaviour restated: kernels/shifts_kernels.h:402-527
// (shift_backward_kernel_nhwdc), cpu/shifts_cpu.cpp:156-181, :221 (the float forward's output -- hence the gradient that comes
// back -- is NCDHW-contiguous even for a channels-last input).
//
// The saved input and grad_x are NDHWC; the incoming gradient is NDHWC or NCDHW-contiguous.  Every channel has its own shift
// in all three dims.  The 2-D kernel's shape (shiftnd_cl_tiled.hip: a workgroup owns 32 channels x a strip of 16 columns and
// walks down the rows of a band through rings of staged rows with a halo of R = 3 pixels) with the depth shift in the STAGING
// ADDRESS, like cl_tiled_forward<.., ND3>: a workgroup works on ONE plane dz of grad_x; its rings are filled element by
// element, every channel from the planes IT reads --
//   * the input ring holds PAIRS (x[plane0(c)], x[plane1(c)]), plane0 = fold(dz - shift_d(c)), plane1 the + 1 corner plane:
//     one LDS read returns both plane corners of a (row, column) corner, four reads the eight corners of
//     compute_weight_gradients (shifts_kernels.h:132-154; corner order :58-103: bit 0 = + 1 along the first spatial dim);
//   * the gradient ring holds, per channel, the one plane the sparse shift's grad_x copies from (grad_out at o + shift,
//     :314-324) -- or, interpolating, the PLANE BLEND lerp(g[p0], g[p1], d_plane) in fp32: the reference nests the blend of
//     the first spatial dim innermost (interpolation.h:34-40), so what is left behind the staging is the 2-D blend, the same
//     bits as interp3D (:287-313);
//   * the gradient at the output position itself (:271) -- plane dz - LD for every channel -- is staged as one more row.
// The workgroups of neighbouring planes (the plane is the second-fastest block index after the channel block) read the same
// pixel lines, each taking the channels whose depth shift points there, and meet in one XCD's L2.
// Weight gradients: the multilinear form (shiftnd_common.hpp: corner_diffs / blend_diffs) -- eight sums of g x corner
// difference per thread (fp32 within a row, fp64 across rows), blended once per thread, combined per workgroup in a fixed
// order into [group][C][3] partial sums that reduce_weight_grads finishes: deterministic, no atomics.
// Shifts beyond the ring (|row / column shift| > 3), the reflected corner one step beyond it and the rows / columns whose
// periodic source wraps are written by an element-by-element pass at the end, as in the 2-D kernel.
// Roofline: HBM, 3 x s bytes per element.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kR = 3;                      // ring half depth = largest |row / column shift| served from LDS
constexpr int kCB = 32;                    // channels per workgroup (fp32: a 128-byte pixel line; 16-bit: half a line)
constexpr int kTW3 = 16;                   // grad_x columns per workgroup
constexpr int kPW3 = kTW3 + 2 * kR + 1;    // staged pixels per row (23)
constexpr int kRing3 = 8;                  // staged rows h - R .. h + R + 1
constexpr int kPL = kThreads / kCB;        // pixel lanes (8)
constexpr int kNI = kTW3 / kPL;            // pixels per thread and row (2)
constexpr int kNS = (kPW3 * kCB + kThreads - 1) / kThreads;   // staged elements per thread, row and plane (3)
constexpr int kOE = kTW3 * kCB / kThreads;                     // own-row elements per thread (NCDHW gradient: 2)
constexpr uint32_t kOutOfRange = 0x80000000u;   // buffer offset beyond every image (num_records < 2^31): loads give 0, stores are dropped
constexpr int kBufferFlags = 0x00020000;        // raw buffer, 32-bit data format (gfx9 family resource word 3)
#ifndef CLT3_DEPTH
#define CLT3_DEPTH 3
#endif

struct ClTiled3Params {
    const char *x, *go;
    char *gx;
    const void *w;
    double *partials;    // [N * bands * wtiles * D][C][3]
    int wkind, N, C, D, H, W, pad;
    int OD, OH, OW, LD, LH, LW;   // the window: grad_out's sizes and its corner in the input volume (no crop: D, H, W, 0, 0, 0)
    int go_ncdhw;                 // the incoming gradient is NCDHW-contiguous (else NDHWC like the saved input and grad_x)
    int wtiles, cblocks, bands, band_rows;
    unsigned xcd_blocks;          // grid / 8 when the XCD-contiguous block remap is on, else 0
    FastDiv d_wtiles, d_cblocks, d_bands, d_D;
    FastDiv d_perD, d_perH, d_perW, d_perOD, d_perOH, d_perOW;
};

template <typename S> __device__ __forceinline__ uint32_t load_elem(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(S) == 4) return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}

// the innermost lerp of interp_t (shiftnd_common.hpp): unfused for fp32 tensors (the reference's bits), fused for 16-bit ones
template <typename T> __device__ __forceinline__ float lerp_t(float a, float b, float x) {
    if constexpr (sizeof(typename T::S) == 2) return lerp1_fused<float>(a, b, x);
    else return lerp1<float>(a, b, x);
}

#ifdef CLT3_WAVES
#define CLT3_OCC __attribute__((amdgpu_waves_per_eu(CLT3_WAVES, CLT3_WAVES)))
#else
#define CLT3_OCC
#endif
template <typename T, bool ACTIVE, bool GO_NCDHW>
__global__ __launch_bounds__(kThreads) CLT3_OCC void cl_tiled_backward_3d(const ClTiled3Params p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    using S = typename T::S;
    using CT = typename T::C;
    static_assert(sizeof(S) == 4 || sizeof(S) == 2, "fp32, fp16, bf16");
    static_assert(sizeof(CT) == 4, "fp32 compute type");
    constexpr int ES = sizeof(S);
    constexpr int PE = 2 * ES;                      // an input-ring element: the pair (plane0, plane1)
    constexpr int GE = ACTIVE ? 4 : ES;             // a gradient-ring element: the fp32 plane blend / the raw element
    // bytes per staged pixel.  Reads have their lanes along the CHANNELS of per-lane different pixels: the natural pitch is
    // conflict-free.  The NCDHW gradient is STAGED with the lanes along the pixels of one channel: one word more per pixel.
    constexpr int XPITCH = kCB * PE, GPITCH = kCB * GE + (GO_NCDHW ? 4 : 0), OPITCH = kCB * ES + (GO_NCDHW ? 4 : 0);
    constexpr int XROW = kPW3 * XPITCH, GROW = kPW3 * GPITCH;
    constexpr int XBASE = 16, GBASE = XBASE + kRing3 * XROW, OBASE = GBASE + kRing3 * GROW, DUMP = OBASE + kTW3 * OPITCH;
    // 16 zero bytes (what a padding tap reads: offsets of taps that do not exist are hugely negative, the address is
    // max(offset sum, 0) -- no lane masks in the row loop), input ring, gradient ring, the own-gradient row, dump bytes
    __shared__ __attribute__((aligned(16))) char lds[DUMP + 16];
    __shared__ int tab_gp0[GO_NCDHW ? kCB : 1], tab_gp1[GO_NCDHW ? kCB : 1];   // NCDHW gradient: the channels' gradient planes
    __shared__ float tab_dd[GO_NCDHW ? kCB : 1];                                // ... and plane fractions (staging lanes run along the pixels)
    constexpr int kNeg = -(1 << 24);
    if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(lds)[threadIdx.x] = 0u;   // (read after the first barrier of the row loop)

    // ---- which tile: channel block fastest, then the plane (neighbouring planes share their pixel lines in one L2) ----------
    unsigned b = p.xcd_blocks ? (blockIdx.x & 7u) * p.xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    const int cb = static_cast<int>(b - fdiv(b, p.d_cblocks) * p.cblocks);
    b = fdiv(b, p.d_cblocks);
    const int pidx = static_cast<int>(b);   // (n, band, wt, dz): the partial-sum group
    const int dz = static_cast<int>(b - fdiv(b, p.d_D) * static_cast<unsigned>(p.D));
    b = fdiv(b, p.d_D);
    const int wt = static_cast<int>(b - fdiv(b, p.d_wtiles) * p.wtiles);
    b = fdiv(b, p.d_wtiles);
    const int band = static_cast<int>(b - fdiv(b, p.d_bands) * p.bands);
    const int n = static_cast<int>(fdiv(b, p.d_bands));
    const int w0 = wt * kTW3, c0 = cb * kCB;
    const int h0 = band * p.band_rows, h1 = min(p.H, h0 + p.band_rows);
    const int D = p.D, H = p.H, W = p.W, C = p.C, OD = p.OD, OH = p.OH, OW = p.OW, LD = p.LD, LH = p.LH, LW = p.LW;
    const uint32_t plane_bytes = static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * static_cast<uint32_t>(C) * ES;
    const uint32_t img_bytes = static_cast<uint32_t>(D) * plane_bytes;   // < 2^31 (host)
    const uint32_t gplane_cl = static_cast<uint32_t>(OH) * static_cast<uint32_t>(OW) * static_cast<uint32_t>(C) * ES;   // a grad_out plane, NDHWC
    const uint32_t gplane_nc = static_cast<uint32_t>(OH) * static_cast<uint32_t>(OW) * ES;                               // ... of one channel, NCDHW
    const uint32_t go_bytes = static_cast<uint32_t>(OD) * gplane_cl;
    const char *xn = p.x + static_cast<int64_t>(n) * img_bytes, *gn = p.go + static_cast<int64_t>(n) * go_bytes;
    char *on = p.gx + static_cast<int64_t>(n) * img_bytes;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xn), 0, img_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gn), 0, go_bytes, kBufferFlags);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(on, 0, img_bytes, kBufferFlags);

    // ---- the thread's channel: shifts, fractions, planes ------------------------------------------------------------------
    const int lane_a = static_cast<int>(threadIdx.x) % kCB, lane_b = static_cast<int>(threadIdx.x) / kCB;
    const int c = c0 + lane_a;
    const bool live_c = c < C;
    int64_t sh[3];
    CT dw[3];   // fractions of the plane, row and column dims (weight columns 0, 1, 2)
    {
        const int wcol[3] = {0, 1, 2};
        CT wv[3];
        load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(live_c ? c : C - 1) * 3, wcol, wv);
#pragma unroll
        for (int k = 0; k < 3; ++k) prep_shift_backward<CT>(wv[k], ACTIVE, sh[k], dw[k]);
    }
    const int csxD = canon_shift(sh[0], D, p.pad, p.d_perD), csxH = canon_shift(sh[1], H, p.pad, p.d_perH), csxW = canon_shift(sh[2], W, p.pad, p.d_perW);
    // grad_x source: the sparse shift reads grad_out at o + shift, the active one at o - shift (shifts_kernels.h:287-293); the
    // gradient's maps fold in the window's sizes (:295-297, :319-324)
    const int csgD = canon_shift(ACTIVE ? sh[0] : -sh[0], OD, p.pad, p.d_perOD), csgH = canon_shift(ACTIVE ? sh[1] : -sh[1], OH, p.pad, p.d_perOH),
              csgW = canon_shift(ACTIVE ? sh[2] : -sh[2], OW, p.pad, p.d_perOW);
    // the channel's planes (size-1 dims ignore the shift, shifts_kernels.h:40-48); -1: padding
    const int xp0 = D == 1 ? 0 : fold_index(dz - csxD, D, p.pad), xp1 = D == 1 ? 0 : fold_index(dz - csxD + 1, D, p.pad);
    const int dzo = dz - LD;                              // this plane in grad_out's coordinates
    const bool ppass = dzo >= 0 && dzo < OD;              // ... lies in the window (else: zero gradient, nothing counted)
    const int gp0 = !ppass ? -1 : (OD == 1 ? 0 : fold_index(dzo - csgD, OD, p.pad));
    const int gp1 = (!ppass || !ACTIVE) ? -1 : (OD == 1 ? 0 : fold_index(dzo - csgD + 1, OD, p.pad));
    if constexpr (GO_NCDHW) {
        if (threadIdx.x < kCB) {
            tab_gp0[threadIdx.x] = gp0;
            tab_gp1[threadIdx.x] = gp1;
            tab_dd[threadIdx.x] = dw[0];
        }
        __syncthreads();
    }
    const int perH = map_period(H, p.pad), perW = map_period(W, p.pad), perOH = map_period(OH, p.pad), perOW = map_period(OW, p.pad);
    const int sh_s = (perH && 2 * csxH > perH) ? csxH - perH : csxH, sw_s = (perW && 2 * csxW > perW) ? csxW - perW : csxW;
    const int gh_s = (perOH && 2 * csgH > perOH) ? csgH - perOH : csgH, gw_s = (perOW && 2 * csgW > perOW) ? csgW - perOW : csgW;
    const bool in_ring = sh_s >= -kR && sh_s <= kR && sw_s >= -kR && sw_s <= kR && gh_s >= -kR && gh_s <= kR && gw_s >= -kR && gw_s <= kR;
    const bool near_c = live_c && in_ring, far_c = live_c && !in_ring;

    // ---- staging: kNS elements per thread, row and plane; element q = k * 256 + tid: pixel q / 32 of the staged row, the
    // thread's own channel (input ring; NDHWC gradient).  NCDHW gradient: element e: channel e / 23, pixel e % 23. ---------
    uint32_t xoff0[kNS], xoff1[kNS], goff0[kNS], goff1[ACTIVE ? kNS : 1];
    int xdst[kNS], gdst[kNS];
    float gdd[GO_NCDHW ? kNS : 1];   // NCDHW gradient: the plane fraction of the staged element's channel
#pragma unroll
    for (int k = 0; k < kNS; ++k) {
        const int q = k * kThreads + static_cast<int>(threadIdx.x);
        const int px = q / kCB, gxs = w0 - kR + px;
        const bool okx = px < kPW3 && gxs >= 0 && gxs < W && live_c;
        const uint32_t inplane = (static_cast<uint32_t>(gxs) * C + c) * ES;
        xoff0[k] = (okx && xp0 >= 0) ? static_cast<uint32_t>(xp0) * plane_bytes + inplane : kOutOfRange;
        xoff1[k] = (okx && xp1 >= 0) ? static_cast<uint32_t>(xp1) * plane_bytes + inplane : kOutOfRange;
        xdst[k] = px < kPW3 ? XBASE + px * XPITCH + lane_a * PE : DUMP;
        if constexpr (GO_NCDHW) {
            const int ch = q / kPW3, pxe = q - ch * kPW3, gxe = w0 - kR + pxe;
            const bool okg = q < kCB * kPW3 && gxe >= LW && gxe < LW + OW && c0 + ch < C;
            const int chs = min(ch, kCB - 1), e0 = tab_gp0[chs], e1 = tab_gp1[chs];
            const uint32_t chan = static_cast<uint32_t>(c0 + ch) * OD;
            goff0[k] = (okg && e0 >= 0) ? (chan + e0) * gplane_nc + static_cast<uint32_t>(gxe - LW) * ES : kOutOfRange;
            if constexpr (ACTIVE) goff1[k] = (okg && e1 >= 0) ? (chan + e1) * gplane_nc + static_cast<uint32_t>(gxe - LW) * ES : kOutOfRange;
            gdd[k] = tab_dd[chs];
            gdst[k] = q < kCB * kPW3 ? GBASE + pxe * GPITCH + ch * GE : DUMP;
        } else {
            const bool okg = px < kPW3 && gxs >= LW && gxs < LW + OW && live_c;
            const uint32_t ing = (static_cast<uint32_t>(gxs - LW) * C + c) * ES;
            goff0[k] = (okg && gp0 >= 0) ? static_cast<uint32_t>(gp0) * gplane_cl + ing : kOutOfRange;
            if constexpr (ACTIVE) goff1[k] = (okg && gp1 >= 0) ? static_cast<uint32_t>(gp1) * gplane_cl + ing : kOutOfRange;
            gdst[k] = px < kPW3 ? GBASE + px * GPITCH + lane_a * GE : DUMP;
        }
    }
    // the own-gradient row (plane dz - LD, row h - LH, the strip's 16 columns x 32 channels).  NDHWC: 16-byte pieces (32 ES of
    // them, one per thread of the first waves); NCDHW: kOE elements per thread, lanes along the pixels
    constexpr int kOPieces = kTW3 * kCB * ES / 16, kOPP = kCB * ES / 16;   // pieces per row / per pixel
    uint32_t ooffp = kOutOfRange, ooffe[GO_NCDHW ? kOE : 1];
    int odstp = DUMP, odste[GO_NCDHW ? kOE : 1];
    if constexpr (GO_NCDHW) {
#pragma unroll
        for (int k = 0; k < kOE; ++k) {
            const int e = k * kThreads + static_cast<int>(threadIdx.x), ch = e / kTW3, pxe = e - ch * kTW3, col = w0 + pxe;
            const bool ok = ppass && col >= LW && col < LW + OW && col < W && c0 + ch < C;
            ooffe[k] = ok ? (static_cast<uint32_t>(c0 + ch) * OD + dzo) * gplane_nc + static_cast<uint32_t>(col - LW) * ES : kOutOfRange;
            odste[k] = OBASE + pxe * OPITCH + ch * ES;
        }
    } else {
        const int q = static_cast<int>(threadIdx.x), pxe = q / kOPP, part = q - pxe * kOPP, col = w0 + pxe;
        const bool ok = q < kOPieces && ppass && col >= LW && col < LW + OW && col < W && (c0 * ES + part * 16) < C * ES;
        ooffp = ok ? static_cast<uint32_t>(dzo) * gplane_cl + (static_cast<uint32_t>(col - LW) * C + c0) * ES + part * 16 : kOutOfRange;
        odstp = q < kOPieces ? OBASE + pxe * OPITCH + part * 16 : DUMP;
    }
    const uint32_t row_bytes = static_cast<uint32_t>(W) * C * ES;
    const uint32_t grow_bytes = static_cast<uint32_t>(OW) * (GO_NCDHW ? 1 : C) * ES;
    constexpr int kDepth = CLT3_DEPTH;
    struct Row {   // one staged row in flight: the thread's elements of both input planes and of the gradient plane(s)
        uint32_t x0[kNS], x1[kNS], g0[kNS], g1[ACTIVE ? kNS : 1];
    };
    struct Own {
        u4 v;
        uint32_t e[GO_NCDHW ? kOE : 1];
    };
    auto load_row = [&](int y, int ylast, Row &r) {   // rows outside the volume or beyond what the band needs: nothing is read
        const int unwanted = (y >> 31) | ((ylast - y) >> 31);
        const uint32_t so = static_cast<uint32_t>(y & ~unwanted) * row_bytes, dead = static_cast<uint32_t>(unwanted) & kOutOfRange;
        const int unwanted_g = unwanted | ((y - LH) >> 31) | ((LH + OH - 1 - y) >> 31);   // grad_out's row y - LH
        const uint32_t sg = static_cast<uint32_t>((y - LH) & ~unwanted_g) * grow_bytes, dead_g = static_cast<uint32_t>(unwanted_g) & kOutOfRange;
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            r.x0[k] = load_elem<S>(xres, xoff0[k] | dead, so);
            r.x1[k] = load_elem<S>(xres, xoff1[k] | dead, so);
            r.g0[k] = load_elem<S>(gres, goff0[k] | dead_g, sg);
            if constexpr (ACTIVE) r.g1[k] = load_elem<S>(gres, goff1[k] | dead_g, sg);
        }
    };
    auto load_own = [&](int y, int yend, Own &o) {   // the gradient at the output positions of row y (a row of the band inside the window)
        const int unwanted = ((yend - 1 - y) >> 31) | ((y - LH) >> 31) | ((LH + OH - 1 - y) >> 31);
        const uint32_t sg = static_cast<uint32_t>((y - LH) & ~unwanted) * grow_bytes, dead = static_cast<uint32_t>(unwanted) & kOutOfRange;
        if constexpr (GO_NCDHW) {
#pragma unroll
            for (int k = 0; k < kOE; ++k) o.e[k] = load_elem<S>(gres, ooffe[k] | dead, sg);
        } else {
            o.v = __builtin_amdgcn_raw_buffer_load_b128(gres, ooffp | dead, sg, 0);
        }
    };
    auto bits_to_f = [](uint32_t b) { return widen<T>(__builtin_bit_cast(S, static_cast<typename raw_t<ES>::type>(b))); };
    auto store_row = [&](int y, const Row &r) {
        const int slot = y & (kRing3 - 1);
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            char *dx = lds + (xdst[k] != DUMP ? xdst[k] + slot * XROW : DUMP);
            if constexpr (ES == 4) {
                uint32_t *d2 = reinterpret_cast<uint32_t *>(__builtin_assume_aligned(dx, 8));
                d2[0] = r.x0[k];
                d2[1] = r.x1[k];
            } else {
                *reinterpret_cast<uint32_t *>(__builtin_assume_aligned(dx, 4)) = (r.x0[k] & 0xffffu) | (r.x1[k] << 16);
            }
            char *dg = lds + (gdst[k] != DUMP ? gdst[k] + slot * GROW : DUMP);
            if constexpr (ACTIVE) {   // the plane blend, innermost like the reference's (interpolation.h:34-40)
                const float dd = GO_NCDHW ? gdd[k] : dw[0];
                *reinterpret_cast<float *>(dg) = lerp_t<T>(bits_to_f(r.g0[k]), bits_to_f(r.g1[k]), dd);
            } else if constexpr (ES == 4) {
                *reinterpret_cast<uint32_t *>(dg) = r.g0[k];
            } else {
                *reinterpret_cast<uint16_t *>(dg) = static_cast<uint16_t>(r.g0[k]);
            }
        }
    };
    auto store_own = [&](const Own &o) {
        if constexpr (GO_NCDHW) {
#pragma unroll
            for (int k = 0; k < kOE; ++k) {
                if constexpr (ES == 4) *reinterpret_cast<uint32_t *>(lds + odste[k]) = o.e[k];
                else *reinterpret_cast<uint16_t *>(lds + odste[k]) = static_cast<uint16_t>(o.e[k]);
            }
        } else {
            *reinterpret_cast<u4 *>(__builtin_assume_aligned(lds + odstp, 16)) = o.v;
        }
    };
    const int ylast = min(H - 1, h1 + kR);
    Row pvs[kDepth];
    Own pvo[kDepth];
    // the ring's first rows h0 - R .. h0 + R, in batches of kDepth through the registers of the rows in flight (all seven at once
    // held 63 - 84 registers that the row loop never needs again: 181 - 225 VGPRs), then the rows of the first steps
    for (int r0 = 0; r0 <= 2 * kR; r0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) load_row(r0 + d <= 2 * kR ? h0 - kR + r0 + d : -1, ylast, pvs[d]);
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int y = h0 - kR + r0 + d;
            if (r0 + d <= 2 * kR && y >= 0 && y < H) store_row(y, pvs[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
        load_row(h0 + kR + 1 + d, ylast, pvs[d]);
        load_own(h0 + d, h1, pvo[d]);
    }

    // ---- the thread's two pixels -------------------------------------------------------------------------------------------
    auto fold_h = [&](int idx) { return H == 1 ? 0 : fold_index(idx, H, p.pad); };   // size-1 dims ignore the shift
    auto fold_w = [&](int idx) { return W == 1 ? 0 : fold_index(idx, W, p.pad); };
    // the gradient's maps: grad_out coordinates in, INPUT coordinates out (-1: padding)
    auto fold_gh = [&](int idx) { const int r = OH == 1 ? 0 : fold_index(idx, OH, p.pad); return r < 0 ? -1 : r + LH; };
    auto fold_gw = [&](int idx) { const int r = OW == 1 ? 0 : fold_index(idx, OW, p.pad); return r < 0 ? -1 : r + LW; };
    int xc0[kNI], xc1[kNI], gc0[kNI], gc1[kNI], od[kNI];
    uint32_t ooff[kNI];
    bool live[kNI], scol[kNI], cpass[kNI];
    int lcount[kNI];   // -1: live and in the window (its terms count), else 0
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
        const int col = lane_b + kPL * i, wq = w0 + min(col, W - 1 - w0);
        cpass[i] = ppass && wq >= LW && wq < LW + OW;
        const int a0 = cpass[i] ? fold_w(wq - csxW) : -1, a1 = cpass[i] ? fold_w(wq - csxW + 1) : -1;
        const int b0 = cpass[i] ? fold_gw(wq - LW - csgW) : -1, b1 = (ACTIVE && cpass[i]) ? fold_gw(wq - LW - csgW + 1) : -1;
        auto outside = [&](int sx) { return sx >= 0 && (sx < w0 - kR || sx > w0 + kTW3 + kR); };
        scol[i] = near_c && w0 + col < W && (outside(a0) || outside(a1) || outside(b0) || outside(b1));
        live[i] = near_c && w0 + col < W && !scol[i];
        lcount[i] = (live[i] && cpass[i]) ? -1 : 0;
        auto lds_col = [&](int sx, int pitch, int esz) { return (live[i] && sx >= 0) ? (sx - (w0 - kR)) * pitch + lane_a * esz : kNeg; };
        xc0[i] = lds_col(a0, XPITCH, PE);
        xc1[i] = lds_col(a1, XPITCH, PE);
        gc0[i] = lds_col(b0, GPITCH, GE);
        gc1[i] = lds_col(b1, GPITCH, GE);
        od[i] = OBASE + col * OPITCH + lane_a * ES;
        ooff[i] = live[i] ? (static_cast<uint32_t>((dz * H + h0) * W + w0 + col) * C + c) * ES : kOutOfRange;
    }
    const uint32_t ostep = static_cast<uint32_t>(W) * C * ES;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    auto row_off = [&](int sy, int rowb) { return ((sy & (kRing3 - 1)) * rowb) | ((sy >> 31) & kNeg); };   // (a negative row: padding)
    // source rows with ONE fold of the signed shift (cl_tiled_backward): valid for |shift| <= R and H >= 5 or H == 1 (host)
    const int fm = (p.pad == 3 || p.pad == 4) ? -1 : 0;
    const int fLo = p.pad == 4 ? -1 : 0, fHi = p.pad == 1 ? H - 1 : (p.pad == 3 ? 2 * H - 2 : 2 * H - 1);
    const int zneg = p.pad == 0 ? kNeg : 0;
    auto fold_once = [&](int idx, int nn, int lo_c, int hi_c) {
        const int below = idx >> 31, above = (nn - 1 - idx) >> 31, t = idx & fm;
        const int r = (idx & ~(below | above)) | ((lo_c - t) & below) | ((hi_c - t) & above) | ((below | above) & zneg);
        return nn == 1 ? 0 : r;
    };
    auto fold1 = [&](int idx) { return fold_once(idx, H, fLo, fHi); };
    const int gLo = p.pad == 4 ? -1 : 0, gHi = p.pad == 1 ? OH - 1 : (p.pad == 3 ? 2 * OH - 2 : 2 * OH - 1);
    auto fold1g = [&](int idx) { return fold_once(idx, OH, gLo, gHi) + LH; };   // (padding stays negative: LH < 2^20)
    // the one source row the rings cannot hold: reflect padding, last row, shift -R (its + 1 corner comes back at distance -(R + 1))
    const bool srow = near_c && p.pad == 3 && H > 1 && sh_s == -kR && h1 == H;
    const bool srow_g = ACTIVE && near_c && p.pad == 3 && OH > 1 && gh_s == -kR;
    const bool periodic = p.pad == 2 && H > 1;
    auto wraps = [&](int h) {
        const int g0 = h - LH - gh_s;
        const bool in_rows = h >= LH && h < LH + OH;
        return periodic && near_c && (h - sh_s < 0 || h + 1 - sh_s >= H || (in_rows && OH > 1 && (g0 < 0 || g0 + (ACTIVE ? 1 : 0) >= OH)));
    };
    int xrow1 = fold1(h0 - sh_s);           // the + 1 corner row of step h - 1 is the first corner row of step h
    int grow1 = fold1g(h0 - LH - gh_s);     // ... and so is the active shift's second gradient row
    const CT dw2[2] = {dw[1], dw[2]};
    auto step = [&](int h, Row &rv, Own &ov) {
        __syncthreads();   // everybody is done with the slot that row h + R + 1 replaces (row h - R - 1) and with the own row
        // (always stored: a row past the volume arrives as zeros -- the rings never hold what was not staged by this workgroup)
        store_row(h + kR + 1, rv);
        store_own(ov);
        __syncthreads();
        load_row(h + kDepth + kR + 1, ylast, rv);
        load_own(h + kDepth, h1, ov);
        const uint32_t so = static_cast<uint32_t>(h - h0) * ostep;
        const int xr0 = row_off(xrow1, XROW);
        xrow1 = fold1(h + 1 - sh_s);
        const int xr1 = row_off(xrow1, XROW);
        const bool rpass = h >= LH && h < LH + OH;
        int gr0, gr1 = kNeg;
        if constexpr (ACTIVE) {
            gr0 = rpass ? row_off(grow1, GROW) : kNeg;
            grow1 = fold1g(h + 1 - LH - gh_s);
            gr1 = rpass ? row_off(grow1, GROW) : kNeg;
        } else {
            gr0 = rpass ? row_off(fold1g(h - LH - gh_s), GROW) : kNeg;
        }
        int skip = 0;   // -1: this thread leaves the row to the element pass
        if (p.pad == 3) skip = ((srow && h == H - 1) || (srow_g && h == LH + OH - 1)) ? -1 : 0;
        else if (p.pad == 2) skip = wraps(h) ? -1 : 0;
        const int stepok = (rpass ? -1 : 0) & ~skip;
        S res[kNI];
        CT s[8] = {CT(0), CT(0), CT(0), CT(0), CT(0), CT(0), CT(0), CT(0)};
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            // the eight input corners: a pair per (row, column) corner; v[plane + 2 row + 4 column]
            CT v[8], df[8];
            auto pair = [&](int ro, int co, CT &a, CT &bb) {
                const char *src = lds + max(XBASE + ro + co, 0);
                if constexpr (ES == 4) {
                    const uint2 pr = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(src, 8));
                    a = __builtin_bit_cast(float, pr.x);
                    bb = __builtin_bit_cast(float, pr.y);
                } else {
                    const uint32_t pr = *reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(src, 4));
                    a = bits_to_f(pr & 0xffffu);
                    bb = bits_to_f(pr >> 16);
                }
            };
            pair(xr0, xc0[i], v[0], v[1]);
            pair(xr1, xc0[i], v[2], v[3]);
            pair(xr0, xc1[i], v[4], v[5]);
            pair(xr1, xc1[i], v[6], v[7]);
            const CT graw = widen<T>(*reinterpret_cast<const S *>(lds + od[i]));
            const uint32_t counted = static_cast<uint32_t>(lcount[i] & stepok);
            corner_diffs<3, CT>(v, df);
            // (masked bit by bit, not multiplied by zero: 0 * NaN is not 0, and a mask of lanes would be scalar work)
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += __builtin_bit_cast(CT, __builtin_bit_cast(uint32_t, graw * df[k]) & counted);
            if constexpr (ACTIVE) {
                auto gtap = [&](int ro, int co) { return *reinterpret_cast<const float *>(lds + max(GBASE + ro + co, 0)); };
                const CT u[4] = {gtap(gr0, gc0[i]), gtap(gr1, gc0[i]), gtap(gr0, gc1[i]), gtap(gr1, gc1[i])};
                res[i] = narrow<T>(interp_t<T, 2>(u, dw2));
            } else {
                res[i] = *reinterpret_cast<const S *>(lds + max(GBASE + gr0 + gc0[i], 0));   // pure copy: the bit pattern is kept
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += static_cast<double>(s[k]);   // (the pixels' terms of one row are added in fp32 first)
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            if constexpr (ES == 4) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, res[i]), ores, ooff[i] | (static_cast<uint32_t>(skip) & kOutOfRange), so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, res[i]), ores, ooff[i] | (static_cast<uint32_t>(skip) & kOutOfRange), so, 0);
            }
        }
    };
    int hb = h0;
    for (; hb + kDepth <= h1; hb += kDepth) {   // whole groups: no condition between the steps (exact wait counts)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) step(hb + d, pvs[d], pvo[d]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)
        if (hb + d < h1) step(hb + d, pvs[d], pvo[d]);

    // ---- what the rings could not serve, from memory, element by element.  One element of channel `ch` (canonical row / column shifts,
    // planes, fractions) at row h, column wq of plane dz: grad_x written, the eight weight-gradient terms added to t[] ------------------
    struct FarP {
        int cxH, cxW, cgH, cgW, xp0, xp1, gp0, gp1;
        CT fr[3];
    };
    auto far_element = [&](int ch, int h, int wq, const FarP &P, double (&t)[8]) {
        const S *xe = reinterpret_cast<const S *>(xn) + ch, *ge = reinterpret_cast<const S *>(gn);
        S *oe = reinterpret_cast<S *>(on) + ch;
        auto g_index = [&](int pl, int r, int cc) {
            return GO_NCDHW ? ((static_cast<int64_t>(ch) * OD + pl) * OH + r) * OW + cc : ((static_cast<int64_t>(pl) * OH + r) * OW + cc) * C + ch;
        };
        auto tap = [&](int pl, int r, int cc) { return (pl >= 0 && r >= 0 && cc >= 0) ? widen<T>(xe[((static_cast<int64_t>(pl) * H + r) * W + cc) * C]) : CT(0); };
        auto gtap_s = [&](int pl, int r, int cc) { return (pl >= 0 && r >= 0 && cc >= 0) ? ge[g_index(pl, r, cc)] : narrow<T>(CT(0)); };
        auto gtap = [&](int pl, int r, int cc) { return widen<T>(gtap_s(pl, r, cc)); };
        const int64_t o = ((static_cast<int64_t>(dz) * H + h) * W + wq) * C;
        if (!(ppass && h >= LH && h < LH + OH && wq >= LW && wq < LW + OW)) {   // outside the window
            oe[o] = narrow<T>(CT(0));
            return;
        }
        const int a0 = fold_w(wq - P.cxW), a1 = fold_w(wq - P.cxW + 1);
        const int b0 = fold_gw(wq - LW - P.cgW) - LW, b1 = fold_gw(wq - LW - P.cgW + 1) - LW;   // grad_out columns (< 0: padding)
        const int r0 = fold_h(h - P.cxH), r1 = fold_h(h - P.cxH + 1);
        const int s0 = fold_gh(h - LH - P.cgH) - LH, s1 = fold_gh(h - LH - P.cgH + 1) - LH;   // grad_out rows (< 0: padding)
        const CT v[8] = {tap(P.xp0, r0, a0), tap(P.xp1, r0, a0), tap(P.xp0, r1, a0), tap(P.xp1, r1, a0),
                         tap(P.xp0, r0, a1), tap(P.xp1, r0, a1), tap(P.xp0, r1, a1), tap(P.xp1, r1, a1)};
        CT df[8];
        const CT gval = widen<T>(ge[g_index(dzo, h - LH, wq - LW)]);
        corner_diffs<3, CT>(v, df);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] += static_cast<double>(gval * df[k]);
        S r;
        if constexpr (ACTIVE) {
            const CT u[8] = {gtap(P.gp0, s0, b0), gtap(P.gp1, s0, b0), gtap(P.gp0, s1, b0), gtap(P.gp1, s1, b0),
                             gtap(P.gp0, s0, b1), gtap(P.gp1, s0, b1), gtap(P.gp0, s1, b1), gtap(P.gp1, s1, b1)};
            r = narrow<T>(interp_t<T, 3>(u, P.fr));
        } else {
            r = gtap_s(P.gp0, s0, b0);
        }
        oe[o] = r;
    };
    // (a) the few elements of channels inside the ring (the reflected corner one step beyond it, wrapping rows / columns): their owner
    bool any_scol = false;
#pragma unroll
    for (int i = 0; i < kNI; ++i) any_scol = any_scol || scol[i];
    const bool wrap_rows = periodic && near_c && (h0 <= kR || h1 >= H - kR - 1 || (h0 <= LH + kR && h1 > LH) || (h1 >= LH + OH - kR - 1 && h0 < LH + OH));
    if (near_c && (any_scol || srow || srow_g || wrap_rows)) {
        FarP P;
        P.cxH = csxH; P.cxW = csxW; P.cgH = csgH; P.cgW = csgW; P.xp0 = xp0; P.xp1 = xp1; P.gp0 = gp0; P.gp1 = gp1;
        P.fr[0] = dw[0]; P.fr[1] = dw[1]; P.fr[2] = dw[2];
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            const int wq = w0 + lane_b + kPL * i;
            if (wq >= W) continue;
            for (int h = h0; h < h1; ++h) {
                if (!(scol[i] || (srow && h == H - 1) || (srow_g && h == LH + OH - 1) || wraps(h))) continue;
                far_element(c, h, wq, P, acc);
            }
        }
    }
    // (b) channels whose row / column shift leaves the ring: all threads of the workgroup share the band's elements of such a channel
    // (its own eight pixel lanes alone would walk the band's rows one after the other: dozens of dependent memory round trips)
    {
        __shared__ unsigned int far_mask_s;
        __syncthreads();
        if (threadIdx.x < 64) {   // (the channel lanes with pixel lane 0 are threads 0 .. 31)
            const unsigned long long m = __ballot(far_c && static_cast<int>(threadIdx.x) < kCB);
            if (threadIdx.x == 0) far_mask_s = static_cast<unsigned int>(m);
        }
        __syncthreads();
        unsigned int fm_left = far_mask_s;
        double *fred = reinterpret_cast<double *>(lds);   // the rings are dead
        while (fm_left) {   // (uniform)
            const int fch = __builtin_ctz(fm_left);
            fm_left &= fm_left - 1;
            const int cf = c0 + fch;
            FarP P;
            {
                const int wcol[3] = {0, 1, 2};
                CT wv[3];
                int64_t fs[3];
                load_weights3<CT>(p.w, p.wkind, static_cast<int64_t>(cf) * 3, wcol, wv);
#pragma unroll
                for (int k = 0; k < 3; ++k) prep_shift_backward<CT>(wv[k], ACTIVE, fs[k], P.fr[k]);
                const int fxD = canon_shift(fs[0], D, p.pad, p.d_perD), fgD = canon_shift(ACTIVE ? fs[0] : -fs[0], OD, p.pad, p.d_perOD);
                P.cxH = canon_shift(fs[1], H, p.pad, p.d_perH);
                P.cxW = canon_shift(fs[2], W, p.pad, p.d_perW);
                P.cgH = canon_shift(ACTIVE ? fs[1] : -fs[1], OH, p.pad, p.d_perOH);
                P.cgW = canon_shift(ACTIVE ? fs[2] : -fs[2], OW, p.pad, p.d_perOW);
                P.xp0 = D == 1 ? 0 : fold_index(dz - fxD, D, p.pad);
                P.xp1 = D == 1 ? 0 : fold_index(dz - fxD + 1, D, p.pad);
                P.gp0 = !ppass ? -1 : (OD == 1 ? 0 : fold_index(dzo - fgD, OD, p.pad));
                P.gp1 = (!ppass || !ACTIVE) ? -1 : (OD == 1 ? 0 : fold_index(dzo - fgD + 1, OD, p.pad));
            }
            double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            const int nelem = (h1 - h0) * kTW3;
            for (int q = static_cast<int>(threadIdx.x); q < nelem; q += kThreads) {
                const int h = h0 + q / kTW3, wq = w0 + q % kTW3;
                if (wq < W) far_element(cf, h, wq, P, t);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) fred[threadIdx.x * 8 + k] = t[k];
            __syncthreads();
            if (static_cast<int>(threadIdx.x) == fch) {   // the channel's owner (pixel lane 0): the threads' sums in thread order
                double u[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for (int k = 0; k < kThreads; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] += fred[k * 8 + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += u[j];
            }
        }
    }

    // ---- the workgroup's weight-gradient partial sums: blended per thread, the pixel lanes of each channel in lane order -----
    double g3[3];
    {
        const double dd[3] = {static_cast<double>(dw[0]), static_cast<double>(dw[1]), static_cast<double>(dw[2])};
        blend_diffs<3>(acc, dd, g3);
    }
    __syncthreads();
    double *red = reinterpret_cast<double *>(lds);   // the rings are dead
#pragma unroll
    for (int k = 0; k < 3; ++k) red[threadIdx.x * 3 + k] = g3[k];
    __syncthreads();
    if (lane_b == 0 && live_c) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < kPL; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) t[j] += red[(k * kCB + lane_a) * 3 + j];
        double *dst = p.partials + (static_cast<size_t>(pidx) * C + c) * 3;
        dst[0] = t[0];
        dst[1] = t[1];
        dst[2] = t[2];
    }
}

bool dense_ndhwc(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[1] == 1 && st[4] == g.C && (sz[1] == 1 || st[3] == g.C * sz[2]) && (sz[0] == 1 || st[2] == g.C * sz[1] * sz[2]) &&
           (g.N == 1 || st[0] == g.C * sz[0] * sz[1] * sz[2]);
}
bool contiguous_ncdhw(const int64_t st[5], const Geometry &g, const int64_t sz[3]) {
    return st[4] == 1 && (sz[1] == 1 || st[3] == sz[2]) && (sz[0] == 1 || st[2] == sz[1] * sz[2]) && st[1] == sz[0] * sz[1] * sz[2] &&
           (g.N == 1 || st[0] == g.C * sz[0] * sz[1] * sz[2]);
}

struct Plan3 {
    int wtiles, cblocks, bands, band_rows;
    int64_t groups;
};
// geometry only (no dtype, no knob): the workspace size must not depend on who asks
Plan3 plan3(const Geometry &g) {
    Plan3 pl;
    const int H = static_cast<int>(g.S[1]), W = static_cast<int>(g.S[2]);
    pl.wtiles = (W + kTW3 - 1) / kTW3;
    pl.cblocks = static_cast<int>((g.C + kCB - 1) / kCB);
    // bands along H: ~7 workgroups per workgroup slot of the chip, at least 8 R rows per band (the ring warm-up is 2 R rows), and
    // (when the batch allows) at most 4096 partial-sum groups per channel
    const int64_t base = g.N * g.S[0] * pl.wtiles;
    int64_t bands = (7168 + base * pl.cblocks - 1) / (base * pl.cblocks);
    const int64_t max_bands = H / (8 * kR) > 0 ? H / (8 * kR) : 1;
    if (bands > max_bands) bands = max_bands;
    while (bands > 1 && base * bands > 4096) --bands;
    if (bands < 1) bands = 1;
    pl.band_rows = static_cast<int>((H + bands - 1) / bands);
    pl.bands = (H + pl.band_rows - 1) / pl.band_rows;
    pl.groups = base * pl.bands;
    return pl;
}

}  // namespace

// geometry gates shared by the eligibility test and the workspace query (the query used to reserve groups * C * 24 bytes
// for EVERY 3-D problem -- contiguous ones, volumes the kernel never takes, and flat volumes (H == 1) whose records came to 0.75 x the
// tensor): no dtype, no strides, no knob -- both callers see the same answer
static bool cl3_geometry_ok(const Geometry &g) {
    if (g.nd != 3 || g.N < 1 || g.C < 1 || g.C % 4 != 0) return false;   // (pixel lines of whole 16-byte pieces: C * es % 16, es >= 2 ... 4)
    for (int d = 0; d < 3; ++d)
        if (g.O[d] < 1 || g.S[d] < 1 || g.L[d] < 0 || g.L[d] + g.O[d] > g.S[d]) return false;
    if (g.S[0] >= (1 << 20) || g.S[1] >= (1 << 20) || g.S[2] >= (1 << 20) || g.N >= (1LL << 24) || g.C >= (1 << 24)) return false;
    if ((g.S[1] != 1 && g.S[1] < 5) || (g.O[1] != 1 && g.O[1] < 5)) return false;   // the kernel folds source rows once
    if (g.C * g.S[0] * g.S[1] * g.S[2] * 2 >= (1LL << 31)) return false;             // (never eligible: 2-byte elements are the smallest)
    const Plan3 pl = plan3(g);
    if (pl.groups * pl.cblocks >= (1LL << 31)) return false;
    // partial-sum records of at most 1/8 of a 2-byte tensor (flat volumes would need 24 / (16 H es) of it)
    return pl.groups * g.C * 24 * 8 <= g.N * g.C * g.S[0] * g.S[1] * g.S[2] * 2 || pl.groups * g.C * 24 <= (1 << 20);
}

// 3-D fp32 / fp16 / bf16; saved input and grad_x dense NDHWC, the incoming gradient NDHWC too or NCDHW-contiguous (a window
// included: grad_out has its sizes); pixel lines of whole 16-byte pieces
bool cl_tiled3_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx) {
    if (g.nd != 3 || (dtype != SHIFTND_F32 && dtype != SHIFTND_F16 && dtype != SHIFTND_BF16)) return false;
    const int es = dtype_size(dtype);
    if (!cl3_geometry_ok(g) || (g.C * es) % 16 != 0) return false;
    if (g.C * g.S[0] * g.S[1] * g.S[2] * es >= (1LL << 31) || g.C * g.O[0] * g.O[1] * g.O[2] * es >= (1LL << 31)) return false;
    if (!dense_ndhwc(g.xs, g, g.S) || !dense_ndhwc(g.gs, g, g.S)) return false;
    const bool go_cl = dense_ndhwc(g.os, g, g.O);
    if (!go_cl && !contiguous_ncdhw(g.os, g, g.O)) return false;
    if (reinterpret_cast<uintptr_t>(x) % es != 0 || reinterpret_cast<uintptr_t>(go) % (go_cl ? 16 : es) != 0 || reinterpret_cast<uintptr_t>(gx) % es != 0) return false;
    return true;
}

size_t cl_tiled3_backward_workspace(const Geometry &g) {
    if (!cl3_geometry_ok(g)) return 0;   // (never eligible: nothing to reserve)
    return static_cast<size_t>(plan3(g).groups) * static_cast<size_t>(g.C) * 3 * sizeof(double);
}

namespace {
template <typename T>
void launch3(const ClTiled3Params &p, const Plan3 &pl, bool active, void *gw, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(pl.groups * pl.cblocks)), block(kThreads);
    if (p.go_ncdhw) {
        if (active) hipLaunchKernelGGL((cl_tiled_backward_3d<T, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((cl_tiled_backward_3d<T, false, true>), grid, block, 0, st, p);
    } else {
        if (active) hipLaunchKernelGGL((cl_tiled_backward_3d<T, true, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((cl_tiled_backward_3d<T, false, false>), grid, block, 0, st, p);
    }
    reduce_weight_grads_of<T>(p.partials, static_cast<int>(pl.groups), p.C, 3, gw, st);
}
}  // namespace

int cl_tiled3_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                       hipStream_t st) {
    const Plan3 pl = plan3(g);
    ClTiled3Params p{};
    p.x = static_cast<const char *>(x);
    p.go = static_cast<const char *>(go);
    p.gx = static_cast<char *>(gx);
    p.w = w;
    p.wkind = dtype;
    p.partials = static_cast<double *>(workspace);
    p.N = static_cast<int>(g.N);
    p.C = static_cast<int>(g.C);
    p.D = static_cast<int>(g.S[0]);
    p.H = static_cast<int>(g.S[1]);
    p.W = static_cast<int>(g.S[2]);
    p.OD = static_cast<int>(g.O[0]);
    p.OH = static_cast<int>(g.O[1]);
    p.OW = static_cast<int>(g.O[2]);
    p.LD = static_cast<int>(g.L[0]);
    p.LH = static_cast<int>(g.L[1]);
    p.LW = static_cast<int>(g.L[2]);
    p.pad = g.pad;
    p.go_ncdhw = dense_ndhwc(g.os, g, g.O) ? 0 : 1;   // (a tensor that is both -- C == 1 -- reads the same either way)
    p.wtiles = pl.wtiles;
    p.cblocks = pl.cblocks;
    p.bands = pl.bands;
    p.band_rows = pl.band_rows;
    p.d_wtiles = make_fastdiv(static_cast<uint32_t>(p.wtiles));
    p.d_cblocks = make_fastdiv(static_cast<uint32_t>(p.cblocks));
    p.d_bands = make_fastdiv(static_cast<uint32_t>(p.bands));
    p.d_D = make_fastdiv(static_cast<uint32_t>(p.D));
    p.d_perD = make_fastdiv(static_cast<uint32_t>(map_period(p.D, p.pad)));
    p.d_perH = make_fastdiv(static_cast<uint32_t>(map_period(p.H, p.pad)));
    p.d_perW = make_fastdiv(static_cast<uint32_t>(map_period(p.W, p.pad)));
    p.d_perOD = make_fastdiv(static_cast<uint32_t>(map_period(p.OD, p.pad)));
    p.d_perOH = make_fastdiv(static_cast<uint32_t>(map_period(p.OH, p.pad)));
    p.d_perOW = make_fastdiv(static_cast<uint32_t>(map_period(p.OW, p.pad)));
    {
        const int64_t grid = pl.groups * pl.cblocks;
        p.xcd_blocks = grid % 8 == 0 ? static_cast<unsigned>(grid / 8) : 0;
    }
    note_kernel(p.go_ncdhw ? "cl_tiled_backward_3d_ncdhw_grad" : "cl_tiled_backward_3d");
    switch (dtype) {
    case SHIFTND_F32: launch3<f32_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_F16: launch3<f16_t>(p, pl, g.active != 0, gw, st); break;
    case SHIFTND_BF16: launch3<bf16_t>(p, pl, g.active != 0, gw, st); break;
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
    return SHIFTND_OK;
}

}  // namespace shiftnd
