// shiftnd_strided.hip -- generic strided fallback kernels (gfx950).
//
// One thread per element, arbitrary element strides (channels-last inputs, non-contiguous
// grads, rows that are not a whole number of 16-byte chunks, dims too large for the LDS maps).
// Index math is 64-bit and evaluates the padding map per element: correct for every input the
// reference accepts, not tuned.  The tuned path is shiftnd_plane.hip.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/):
//   forward   kernels/shifts_kernels.h:156-220 + cpu/shifts_cpu.cpp:216-232
//   backward  kernels/shifts_kernels.h:222-327 + cpu/shifts_cpu.cpp:237-255
//   quantized kernels/shifts_kernels.h:532-571 + quantized/shifts_quantized.cpp:107-130
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
// the final stage of every family's weight-gradient reduction (declared in shiftnd_common.hpp)
namespace {
template <typename T>
__global__ __launch_bounds__(64) void reduce_weight_grads(const double *__restrict__ partials, int groups, int C, int nd,
                                                           typename T::S *__restrict__ grad_w) {
    // one wave per output element: lanes stride over the groups (fixed order), then a fixed shuffle tree
    const int t = blockIdx.x;
    const int c = t / nd, s = t - c * nd;
    double acc = 0.0;
    for (int g = threadIdx.x; g < groups; g += 64) acc += partials[(static_cast<size_t>(g) * C + c) * 3 + s];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        if constexpr (sizeof(typename T::S) == 8) {
            grad_w[t] = acc;
        } else {
            grad_w[t] = narrow<T>(static_cast<float>(acc));
        }
    }
}
}  // namespace

void launch_reduce_weight_grads(int dtype, const double *partials, int groups, int C, int nd, void *grad_w, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(C) * nd), block(64);
    switch (dtype) {
    case SHIFTND_F32: hipLaunchKernelGGL((reduce_weight_grads<f32_t>), grid, block, 0, st, partials, groups, C, nd, static_cast<float *>(grad_w)); break;
    case SHIFTND_F64: hipLaunchKernelGGL((reduce_weight_grads<f64_t>), grid, block, 0, st, partials, groups, C, nd, static_cast<double *>(grad_w)); break;
    case SHIFTND_F16: hipLaunchKernelGGL((reduce_weight_grads<f16_t>), grid, block, 0, st, partials, groups, C, nd, static_cast<f16_t::S *>(grad_w)); break;
    default: hipLaunchKernelGGL((reduce_weight_grads<bf16_t>), grid, block, 0, st, partials, groups, C, nd, static_cast<bf16_t::S *>(grad_w)); break;
    }
}

namespace {

// resolve one gather: returns element offset within the (n) slice or -1 (fill)
__device__ __forceinline__ int64_t resolve(const int64_t idx[3], const int64_t size[3], const int64_t *st /*d0,d1,inner*/,
                                           int pad) {
    int64_t off = 0;
#pragma unroll 1
    for (int d = 0; d < 3; ++d) {   // one copy of the padding map (64-bit divisions) in the code
        const int64_t t = (size[d] == 1) ? 0 : pad_index(idx[d], size[d], pad);
        if (t < 0) return -1;
        off += t * st[d];
    }
    return off;
}

// SSL / quantized forward: pure gather of ESIZE-byte elements ------------------------------------
template <int ESIZE>
__global__ __launch_bounds__(kThreads) void strided_gather_forward(Geometry g, const void *__restrict__ x_,
                                                                    const void *__restrict__ w, int wkind, int64_t wzp,
                                                                    uint64_t fill_bits, void *__restrict__ out_,
                                                                    int64_t total) {
    using R = typename raw_t<ESIZE>::type;
    const R *x = static_cast<const R *>(x_);
    R *out = static_cast<R *>(out_);
    const R fill = static_cast<R>(fill_bits);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t r = e;
        const int64_t o2 = r % g.O[2]; r /= g.O[2];
        const int64_t o1 = r % g.O[1]; r /= g.O[1];
        const int64_t o0 = r % g.O[0]; r /= g.O[0];
        const int64_t c = r % g.C;
        const int64_t n = r / g.C;
        const int64_t o[3] = {o0, o1, o2};
        int64_t idx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int64_t sh = g.wcol[d] >= 0 ? gather_shift(w, wkind, wzp, c * g.nd + g.wcol[d]) : 0;
            idx[d] = o[d] + g.L[d] - sh;
        }
        const int64_t off = resolve(idx, g.S, g.xs + 2, g.pad);
        const R v = off >= 0 ? x[n * g.xs[0] + c * g.xs[1] + off] : fill;
        out[n * g.os[0] + c * g.os[1] + o0 * g.os[2] + o1 * g.os[3] + o2 * g.os[4]] = v;
    }
}

// gather the 2^ND corners of `arr` around idx (normalised dims), masked by `pass`.  The padding map is separable: each
// dim resolves its two coordinates once (6 pad_index evaluations, not 3 per corner -- every one carries 64-bit divisions,
// and the all-corners form made these kernels 165-325 KB of code each).
template <typename T, int ND>
__device__ __forceinline__ void gather_corners(const typename T::S *arr, const int64_t idx[3], const int64_t size[3],
                                               const int64_t *st, int pad, bool pass, typename T::C *v) {
    int64_t ax[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};   // element offset of coordinate idx[d] + k along dim d, or -1 (fill)
    // one copy of the padding map in the code, walked 2 * ND + (3 - ND) times (the registers are picked by compile-time
    // selects: a dynamically indexed local array would live in scratch)
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        const int d = j >> 1, k = j & 1;
        if (k == 1 && d < 3 - ND) continue;   // leading (size-1) dims have one coordinate
        const int64_t sz = d == 0 ? size[0] : (d == 1 ? size[1] : size[2]);
        const int64_t at = (d == 0 ? idx[0] : (d == 1 ? idx[1] : idx[2])) + k;
        const int64_t sd = d == 0 ? st[0] : (d == 1 ? st[1] : st[2]);
        const int64_t t = (sz == 1) ? 0 : pad_index(at, sz, pad);
        const int64_t r = t < 0 ? -1 : t * sd;
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) ax[dd][kk] = (j == dd * 2 + kk) ? r : ax[dd][kk];
    }
#pragma unroll
    for (int q = 0; q < (1 << ND); ++q) {
        int64_t off = 0;
        bool in = pass;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int k = d >= 3 - ND ? (q >> (d - (3 - ND))) & 1 : 0;
            in = in && ax[d][k] >= 0;
            off += ax[d][k];
        }
        v[q] = in ? widen<T>(arr[off]) : typename T::C(0);
    }
}

// active forward ----------------------------------------------------------------------------------
template <typename T, int ND>
__global__ __launch_bounds__(kThreads) void strided_active_forward(Geometry g, const typename T::S *__restrict__ x,
                                                                    const typename T::S *__restrict__ w,
                                                                    typename T::S *__restrict__ out, int64_t total) {
    using CT = typename T::C;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t r = e;
        const int64_t o2 = r % g.O[2]; r /= g.O[2];
        const int64_t o1 = r % g.O[1]; r /= g.O[1];
        const int64_t o0 = r % g.O[0]; r /= g.O[0];
        const int64_t c = r % g.C;
        const int64_t n = r / g.C;
        const int64_t o[3] = {o0, o1, o2};
        int64_t idx[3];
        CT dw[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int64_t sh = 0;
            if (g.wcol[d] >= 0) prep_shift_forward<CT>(load_weight<CT>(w, T::kDtype, c * g.nd + g.wcol[d]), true, sh, dw[g.wcol[d]]);
            idx[d] = o[d] + g.L[d] - sh;
        }
        CT v[8];
        gather_corners<T, ND>(x + n * g.xs[0] + c * g.xs[1], idx, g.S, g.xs + 2, g.pad, true, v);
        out[n * g.os[0] + c * g.os[1] + o0 * g.os[2] + o1 * g.os[3] + o2 * g.os[4]] = narrow<T>(interp_t<T, ND>(v, dw));
    }
}

// backward: one workgroup per (n, c) plane ---------------------------------------------------------
template <typename T, int ND, bool ACTIVE>
__global__ __launch_bounds__(kThreads) void strided_backward(Geometry g, const typename T::S *__restrict__ go,
                                                             const typename T::S *__restrict__ x,
                                                             const typename T::S *__restrict__ w,
                                                             typename T::S *__restrict__ gx,
                                                             double *__restrict__ partials) {
    using CT = typename T::C;
    __shared__ double scratch[kThreads / 64];
    const int64_t plane = blockIdx.x;
    const int64_t c = plane % g.C, n = plane / g.C;
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (g.wcol[d] >= 0) prep_shift_backward<CT>(load_weight<CT>(w, T::kDtype, c * g.nd + g.wcol[d]), ACTIVE, sh[d], dw[g.wcol[d]]);

    const typename T::S *xp = x + n * g.xs[0] + c * g.xs[1];
    const typename T::S *gop = go + n * g.os[0] + c * g.os[1];
    typename T::S *gxp = gx + n * g.gs[0] + c * g.gs[1];
    const int64_t elems = g.S[0] * g.S[1] * g.S[2];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t e = threadIdx.x; e < elems; e += kThreads) {
        int64_t r = e;
        const int64_t i2 = r % g.S[2]; r /= g.S[2];
        const int64_t i1 = r % g.S[1];
        const int64_t i0 = r / g.S[1];
        const int64_t in[3] = {i0, i1, i2};
        bool pass = true;
        int64_t o[3], xi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            o[d] = in[d] - g.L[d];
            pass = pass && (o[d] >= 0) && (o[d] < g.O[d]);
            xi[d] = in[d] - sh[d];
        }
        const CT gval = pass ? widen<T>(gop[o[0] * g.os[2] + o[1] * g.os[3] + o[2] * g.os[4]]) : CT(0);
        CT v[8], wg[3];
        gather_corners<T, ND>(xp, xi, g.S, g.xs + 2, g.pad, pass, v);
        weight_grads_nd<ND, CT>(v, dw, wg);
        if (pass) {
#pragma unroll
            for (int s = 0; s < ND; ++s) acc[s] += static_cast<double>(gval * wg[s]);
        }
        CT res = CT(0);
        if (ACTIVE) {
            int64_t gi[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) gi[d] = o[d] - sh[d];
            gather_corners<T, ND>(gop, gi, g.O, g.os + 2, g.pad, pass, v);
            res = pass ? interp_t<T, ND>(v, dw) : CT(0);
            gxp[i0 * g.gs[2] + i1 * g.gs[3] + i2 * g.gs[4]] = narrow<T>(res);
        } else {
            int64_t gi[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) gi[d] = o[d] + sh[d];
            const int64_t off = pass ? resolve(gi, g.O, g.os + 2, g.pad) : -1;
            // pure copy: keep the bit pattern
            typename T::S outv = off >= 0 ? gop[off] : narrow<T>(CT(0));
            gxp[i0 * g.gs[2] + i1 * g.gs[3] + i2 * g.gs[4]] = outv;
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double t = block_sum(acc[s], scratch);
        if (threadIdx.x == 0) partials[plane * 3 + s] = t;
    }
}

template <typename T, int ND>
int launch_strided_backward_nd(const Geometry &g, const void *go, const void *x, const void *w, void *gx, void *gw,
                               double *partials, hipStream_t st) {
    using S = typename T::S;
    const int64_t planes = g.N * g.C;
    if (planes > 0x7fffffffLL) return SHIFTND_ERR_TOO_LARGE;
    if (g.active)
        hipLaunchKernelGGL((strided_backward<T, ND, true>), dim3(static_cast<unsigned>(planes)), dim3(kThreads), 0, st, g,
                           static_cast<const S *>(go), static_cast<const S *>(x), static_cast<const S *>(w),
                           static_cast<S *>(gx), partials);
    else
        hipLaunchKernelGGL((strided_backward<T, ND, false>), dim3(static_cast<unsigned>(planes)), dim3(kThreads), 0, st, g,
                           static_cast<const S *>(go), static_cast<const S *>(x), static_cast<const S *>(w),
                           static_cast<S *>(gx), partials);
    reduce_weight_grads_of<T>(partials, static_cast<int>(g.N), static_cast<int>(g.C), g.nd, gw, st);
    return SHIFTND_OK;
}

template <typename T>
int launch_strided_backward_t(const Geometry &g, const void *go, const void *x, const void *w, void *gx, void *gw,
                              double *partials, hipStream_t st) {
    switch (g.nd) {
    case 1: return launch_strided_backward_nd<T, 1>(g, go, x, w, gx, gw, partials, st);
    case 2: return launch_strided_backward_nd<T, 2>(g, go, x, w, gx, gw, partials, st);
    default: return launch_strided_backward_nd<T, 3>(g, go, x, w, gx, gw, partials, st);
    }
}

template <typename T>
int launch_strided_active_t(const Geometry &g, const void *x, const void *w, void *out, int64_t total, unsigned grid,
                            hipStream_t st) {
    using S = typename T::S;
    switch (g.nd) {
    case 1:
        hipLaunchKernelGGL((strided_active_forward<T, 1>), dim3(grid), dim3(kThreads), 0, st, g,
                           static_cast<const S *>(x), static_cast<const S *>(w), static_cast<S *>(out), total);
        break;
    case 2:
        hipLaunchKernelGGL((strided_active_forward<T, 2>), dim3(grid), dim3(kThreads), 0, st, g,
                           static_cast<const S *>(x), static_cast<const S *>(w), static_cast<S *>(out), total);
        break;
    default:
        hipLaunchKernelGGL((strided_active_forward<T, 3>), dim3(grid), dim3(kThreads), 0, st, g,
                           static_cast<const S *>(x), static_cast<const S *>(w), static_cast<S *>(out), total);
        break;
    }
    return SHIFTND_OK;
}

unsigned flat_grid(int64_t total) {
    const int64_t blocks = (total + kThreads - 1) / kThreads;
    const int64_t cap = 256LL * 32;  // 32 workgroups per CU, grid-stride beyond
    return static_cast<unsigned>(blocks < cap ? blocks : cap);
}

}  // namespace

int strided_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp,
                    uint64_t fill_bits, void *out, hipStream_t st) {
    const int64_t total = g.N * g.C * g.O[0] * g.O[1] * g.O[2];
    const unsigned grid = flat_grid(total);
    const bool gather_only = !g.active || dtype >= SHIFTND_I8;
    note_kernel(gather_only ? "strided_gather_forward" : "strided_active_forward");
    if (gather_only) {
        switch (dtype_size(dtype)) {
        case 1:
            hipLaunchKernelGGL((strided_gather_forward<1>), dim3(grid), dim3(kThreads), 0, st, g, x, w, wkind, wzp,
                               fill_bits, out, total);
            break;
        case 2:
            hipLaunchKernelGGL((strided_gather_forward<2>), dim3(grid), dim3(kThreads), 0, st, g, x, w, wkind, wzp,
                               fill_bits, out, total);
            break;
        case 4:
            hipLaunchKernelGGL((strided_gather_forward<4>), dim3(grid), dim3(kThreads), 0, st, g, x, w, wkind, wzp,
                               fill_bits, out, total);
            break;
        default:
            hipLaunchKernelGGL((strided_gather_forward<8>), dim3(grid), dim3(kThreads), 0, st, g, x, w, wkind, wzp,
                               fill_bits, out, total);
            break;
        }
        return SHIFTND_OK;
    }
    switch (dtype) {
    case SHIFTND_F32: return launch_strided_active_t<f32_t>(g, x, w, out, total, grid, st);
    case SHIFTND_F64: return launch_strided_active_t<f64_t>(g, x, w, out, total, grid, st);
    case SHIFTND_F16: return launch_strided_active_t<f16_t>(g, x, w, out, total, grid, st);
    case SHIFTND_BF16: return launch_strided_active_t<bf16_t>(g, x, w, out, total, grid, st);
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
}

size_t strided_backward_workspace(const Geometry &g) { return static_cast<size_t>(g.N * g.C) * 3 * sizeof(double); }

int strided_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                     void *workspace, hipStream_t st) {
    double *partials = static_cast<double *>(workspace);
    note_kernel("strided_backward");
    switch (dtype) {
    case SHIFTND_F32: return launch_strided_backward_t<f32_t>(g, go, x, w, gx, gw, partials, st);
    case SHIFTND_F64: return launch_strided_backward_t<f64_t>(g, go, x, w, gx, gw, partials, st);
    case SHIFTND_F16: return launch_strided_backward_t<f16_t>(g, go, x, w, gx, gw, partials, st);
    case SHIFTND_BF16: return launch_strided_backward_t<bf16_t>(g, go, x, w, gx, gw, partials, st);
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
}

}  // namespace shiftnd
