// shiftnd_qpool.hip -- quantized shift + average pool in one pass (gfx950 / MI355X): the tail a quantized module that
// emulates a strided depthwise conv attaches to its shift (`_reduction_fn(shift(x))`,
// torchshifts/quantized/modules/shifts.py:19-20 with modules/shifts.py:81-89: avg_pool{N}d(kernel = stride, ceil_mode = True)).
// ATen has no QuantizedCUDA average pool; a composite of float ops on the integer representation takes six passes
// over the tensor (round 2).  Here one thread produces one pooled element: it gathers its window through the
// channel's padding map (arithmetic: canon_shift + fold_index), sums x_int - zero_point, and requantizes exactly like
// ATen's QuantizedCPU kernels (fp32, clamped to the type's range, the scale does not change; ATen rounds with the zero point
// inside -- contiguous 1-D / 2-D -- or outside -- channels-last, 3-D -- the rounding: `requant` picks).  One-byte element types; contiguous tensors; any padding, crop and number of dims.
//
// Reference behaviour restated: kernels/shifts_kernels.h:532-571 (quantized shift: shift = int_repr(w) - w.zero_point,
// fill = x.zero_point); the pool is ATen's.  Roofline: HBM, (1 + 1 / window) bytes per element.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

struct QPoolParams {
    const void *x;
    void *out;
    const void *w;
    int64_t wzp;
    int32_t xzp, qmin, qmax;
    int wkind, C, nd, pad, zp_outside;
    int S[3], O[3], L[3], K[3], P[3], wcol[3];
    int64_t x_plane, p_plane;
    uint32_t chunks;  // workgroups per (n, c) plane of the pooled output
    FastDiv d_chunks, d_C, d_P2, d_P12;
    FastDiv d_per[3];
};

template <typename EL>
__global__ __launch_bounds__(kThreads) void qpool_forward(const QPoolParams p) {
    const uint32_t plane = fdiv(blockIdx.x, p.d_chunks);
    const uint32_t chunk = blockIdx.x - plane * p.chunks;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            cs[d] = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol[d]), p.S[d], p.pad, p.d_per[d]);
    const uint32_t t = chunk * kThreads + threadIdx.x;
    if (t >= static_cast<uint32_t>(p.p_plane)) return;
    const int p0 = static_cast<int>(fdiv(t, p.d_P12));
    const uint32_t rem = t - static_cast<uint32_t>(p0) * static_cast<uint32_t>(p.P[1] * p.P[2]);
    const int p1 = static_cast<int>(fdiv(rem, p.d_P2));
    const int p2 = static_cast<int>(rem) - p1 * p.P[2];
    const int n0 = min(p.K[0], p.O[0] - p0 * p.K[0]), n1 = min(p.K[1], p.O[1] - p1 * p.K[1]), n2 = min(p.K[2], p.O[2] - p2 * p.K[2]);
    const EL *xp = static_cast<const EL *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    int acc = 0;
    for (int a = 0; a < n0; ++a) {
        const int sa = p.S[0] == 1 ? 0 : fold_index(p0 * p.K[0] + a + p.L[0] - cs[0], p.S[0], p.pad);
        for (int b = 0; b < n1; ++b) {
            const int sb = p.S[1] == 1 ? 0 : fold_index(p1 * p.K[1] + b + p.L[1] - cs[1], p.S[1], p.pad);
            const bool rowok = sa >= 0 && sb >= 0;
            const EL *row = xp + static_cast<int64_t>((rowok ? sa : 0) * p.S[1] + (rowok ? sb : 0)) * p.S[2];
            for (int k = 0; k < n2; ++k) {
                const int sc = p.S[2] == 1 ? 0 : fold_index(p2 * p.K[2] + k + p.L[2] - cs[2], p.S[2], p.pad);
                if (rowok && sc >= 0) acc += static_cast<int>(row[sc]) - p.xzp;  // (fill = zero point: contributes 0)
            }
        }
    }
    // ATen's QuantizedCPU average pool: multiplier = float(in_scale / out_scale / count) = float(1 / count).  Its contiguous
    // 1-D / 2-D kernel requantizes through quantize_val(scale = 1 / multiplier) -> fbgemm::Quantize: nearbyint(zero_point +
    // sum * (1 / scale)), the zero point INSIDE the rounding; its channels-last kernel (every 3-D tensor too) rounds sum *
    // multiplier and adds the zero point afterwards.  Both pinned against torch's CPU kernels by
    // tests/test_quant_convert.py::test_quantized_avg_pool_restatement_matches_aten.
    const float mult = static_cast<float>(1.0 / static_cast<double>(n0 * n1 * n2));
    int q;
    if (p.zp_outside) {
        q = static_cast<int>(nearbyintf(static_cast<float>(acc) * mult)) + p.xzp;
    } else {
        const float scale = 1.0f / mult;
        const float inv = 1.0f / scale;
        q = static_cast<int>(nearbyintf(static_cast<float>(p.xzp) + static_cast<float>(acc) * inv));
    }
    q = q < p.qmin ? p.qmin : (q > p.qmax ? p.qmax : q);
    static_cast<EL *>(p.out)[static_cast<int64_t>(plane) * p.p_plane + t] = static_cast<EL>(q);
}

}  // namespace

bool qpool_forward_eligible(const Geometry &g, int dtype) {
    if (dtype != SHIFTND_I8 && dtype != SHIFTND_U8) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], pe = g.P[0] * g.P[1] * g.P[2];
    if (xe < 1 || pe < 1 || xe >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return false;
    const int64_t chunks = (pe + kThreads - 1) / kThreads;
    return g.N * g.C * chunks < (1LL << 31);
}

int qpool_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, int64_t xzp, int requant, void *out,
                  hipStream_t st) {
    QPoolParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wzp = wzp;
    p.xzp = static_cast<int32_t>(xzp);
    p.qmin = dtype == SHIFTND_I8 ? -128 : 0;
    p.qmax = dtype == SHIFTND_I8 ? 127 : 255;
    p.wkind = wkind;
    p.zp_outside = requant == SHIFTND_REQUANT_ZP_OUTSIDE ? 1 : 0;
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.O[d] = static_cast<int>(g.O[d]);
        p.L[d] = static_cast<int>(g.L[d]);
        p.K[d] = static_cast<int>(g.K[d] > 0 ? g.K[d] : 1);
        p.P[d] = static_cast<int>(g.P[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.p_plane = g.P[0] * g.P[1] * g.P[2];
    p.chunks = static_cast<uint32_t>((p.p_plane + kThreads - 1) / kThreads);
    p.d_chunks = make_fastdiv(p.chunks);
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    p.d_P2 = make_fastdiv(static_cast<uint32_t>(g.P[2]));
    p.d_P12 = make_fastdiv(static_cast<uint32_t>(g.P[1] * g.P[2]));
    note_kernel("qpool_forward");
    const dim3 grid(static_cast<unsigned>(g.N * g.C * p.chunks)), block(kThreads);
    if (dtype == SHIFTND_I8) hipLaunchKernelGGL((qpool_forward<int8_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((qpool_forward<uint8_t>), grid, block, 0, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
