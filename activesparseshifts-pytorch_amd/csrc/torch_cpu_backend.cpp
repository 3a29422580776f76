// torch_cpu_backend.cpp -- CPU and QuantizedCPU dispatch keys of the `torchshifts` library.
//
// The reference ships a CPU backend (csrc/ops/cpu/shifts_cpu.cpp, quantized/shifts_quantized.cpp);
// a drop-in must keep serving CPU tensors.  This is a host implementation for CPU TENSORS ONLY --
// it is not a fallback for the HIP path (HIP tensors dispatch to the CUDA key in torch_binding.cpp
// and raise if the HIP library fails) and it does not use anything under oracle/.
//
// Structure: one task per (n, c) plane (at::parallel_for), the padding index map of each spatial
// dim is evaluated once per channel instead of per element, arbitrary strides (NCHW and
// channels-last inputs take the same code), weight-gradient partials are kept per plane in fp64
// and summed over n afterwards -- race-free and deterministic, unlike the reference's shared `+=`
// (global_scope.h:22).
//
// Per-element math restated from kernels/shifts_kernels.h:10-327, :532-571 and interpolation.h.
#include <ATen/ATen.h>
#include <ATen/Parallel.h>
#include <torch/library.h>

#include <cmath>
#include <cstring>
#include <tuple>
#include <vector>

namespace torchshifts_amd {
namespace cpu {

using at::Tensor;

inline int64_t pmod(int64_t a, int64_t b) { return (b + (a % b)) % b; }

// infer_index + validity (shifts_kernels.h:10-29, :40-41): source index or -1 = fill
inline int64_t pad_index(int64_t idx, int64_t len, int pad) {
    int64_t r;
    switch (pad) {
    case 1: r = idx < 0 ? 0 : (idx > len - 1 ? len - 1 : idx); break;
    case 2: r = pmod(idx, len); break;
    case 3: {
        const int64_t neg = idx < 0, a = idx < 0 ? -idx : idx;
        const bool odd = ((neg + (a - neg) / (len - 1)) & 1) != 0;
        const int64_t m = pmod(idx, len - 1);
        r = odd ? len - 1 - m : m;
        break;
    }
    case 4: {
        const int64_t neg = idx < 0, a = idx < 0 ? -idx : idx;
        const bool odd = ((neg + (a - neg) / len) & 1) != 0;
        const int64_t m = pmod(idx, len);
        r = odd ? len - 1 - m : m;
        break;
    }
    default: r = idx > len - 1 ? -1 : idx; break;
    }
    return r < 0 ? -1 : r;
}

struct Geom {
    int nd = 1, pad = 0;
    bool active = false;
    int64_t N = 0, C = 0;
    int64_t S[3] = {1, 1, 1}, O[3] = {1, 1, 1}, L[3] = {0, 0, 0};  // normalised: real dim r -> r + 3 - nd
    int64_t xs[5] = {0, 0, 0, 0, 0}, os[5] = {0, 0, 0, 0, 0}, gs[5] = {0, 0, 0, 0, 0};
    int wcol[3] = {-1, -1, -1};
};

void set_strides(const Tensor &t, int nd, int64_t out[5]) {
    out[0] = t.stride(0);
    out[1] = t.stride(1);
    for (int r = 0; r < nd; ++r) out[2 + r + 3 - nd] = t.stride(2 + r);
}

Geom make_geom(int nd, const Tensor &input, const int32_t b[6], int64_t pad, bool active) {
    Geom g;
    g.nd = nd;
    g.pad = static_cast<int>(pad);
    g.active = active;
    g.N = input.size(0);
    g.C = input.size(1);
    for (int r = 0; r < nd; ++r) {
        const int d = r + 3 - nd;
        g.S[d] = input.size(2 + r);
        g.L[d] = b[2 * r];
        g.O[d] = b[2 * r + 1] - b[2 * r];
        g.wcol[d] = r;
    }
    set_strides(input, nd, g.xs);
    return g;
}

// maps[c][d]: size[d] + 1 entries, map[p] = pad(p + sign * shift[c][d])
struct Maps {
    int64_t off[3], per_channel;
    std::vector<int32_t> data;
    const int32_t *get(int64_t c, int d) const { return data.data() + c * per_channel + off[d]; }
};

Maps build_maps(const Geom &g, const int64_t size[3], const std::vector<int64_t> &shifts /*[C][3] normalised*/, int sign) {
    Maps m;
    m.off[0] = 0;
    m.off[1] = size[0] + 1;
    m.off[2] = m.off[1] + size[1] + 1;
    m.per_channel = m.off[2] + size[2] + 1;
    m.data.resize(static_cast<size_t>(g.C * m.per_channel));
    for (int64_t c = 0; c < g.C; ++c)
        for (int d = 0; d < 3; ++d)
            for (int64_t p = 0; p <= size[d]; ++p)
                m.data[c * m.per_channel + m.off[d] + p] =
                    size[d] == 1 ? 0 : static_cast<int32_t>(pad_index(p + sign * shifts[c * 3 + d], size[d], g.pad));
    return m;
}

template <typename CT> inline CT lerp1(CT v1, CT v2, CT x) { return v1 * (CT(1) - x) + v2 * x; }

template <typename CT> inline CT interp_nd(int nd, const CT *v, const CT *d) {
    if (nd == 1) return lerp1(v[0], v[1], d[0]);
    if (nd == 2) return lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]);
    return lerp1(lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]),
                 lerp1(lerp1(v[4], v[5], d[0]), lerp1(v[6], v[7], d[0]), d[1]), d[2]);
}

template <typename CT> inline void weight_grads_nd(int nd, const CT *v, const CT *d, CT *g) {
    if (nd == 1) {
        g[0] = v[1] - v[0];
    } else if (nd == 2) {
        g[0] = lerp1(v[2] - v[0], v[3] - v[1], d[1]);
        g[1] = lerp1(v[2], v[3], d[0]) - lerp1(v[0], v[1], d[0]);
    } else {
        g[0] = lerp1(lerp1(v[2] - v[0], v[3] - v[1], d[1]), lerp1(v[6] - v[4], v[7] - v[5], d[1]), d[2]);
        g[1] = lerp1(lerp1(v[2], v[3], d[0]) - lerp1(v[0], v[1], d[0]), lerp1(v[6], v[7], d[0]) - lerp1(v[4], v[5], d[0]),
                     d[2]);
        g[2] = lerp1(lerp1(v[4], v[5], d[0]), lerp1(v[6], v[7], d[0]), d[1]) -
               lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]);
    }
}

template <typename T> struct compute_of { using type = float; };
template <> struct compute_of<double> { using type = double; };

// corners of `arr` (plane base) at coordinates p (+1 per corner bit) through the maps of one channel
template <typename T, typename CT>
inline void corners(int nd, const T *arr, const int64_t *st /*d0,d1,inner*/, const int32_t *const mp[3], const int64_t p[3],
                    bool pass, CT *v) {
    for (int q = 0; q < (1 << nd); ++q) {
        int64_t off = 0;
        bool ok = pass;
        for (int d = 0; d < 3 && ok; ++d) {
            const int r = d - (3 - nd);  // real dim of normalised dim d
            const int64_t pp = p[d] + ((r >= 0) ? ((q >> r) & 1) : 0);
            const int32_t m = mp[d][pp];
            ok = m >= 0;
            off += m * st[d];
        }
        v[q] = ok ? static_cast<CT>(arr[off]) : CT(0);
    }
}

// cpu/shifts_cpu.cpp:223-224 / :242-244 (rounding half-to-even = nearbyint in the default mode)
template <typename CT> void prep_forward(CT w, bool active, int64_t &iw, CT &dw) {
    const CT r = active ? std::floor(w) : std::nearbyint(w);
    iw = static_cast<int64_t>(r);
    dw = active ? (w - static_cast<CT>(iw)) : CT(0);
}
template <typename CT> void prep_backward(CT w, bool active, int64_t &iw, CT &dw) {
    dw = active ? (w - std::floor(w)) : (w > CT(0) ? (w - std::floor(w)) : (std::ceil(w) - w));
    const CT r = active ? (w - dw) : std::nearbyint(w);
    iw = static_cast<int64_t>(r);
}

template <typename T> void forward_float(const Geom &g, const Tensor &input, const Tensor &weights, Tensor &output) {
    using CT = typename compute_of<T>::type;
    const T *x = input.data_ptr<T>();
    T *out = output.data_ptr<T>();
    Tensor wc = weights.contiguous();
    const T *w = wc.data_ptr<T>();
    std::vector<int64_t> shifts(static_cast<size_t>(g.C * 3), 0);
    std::vector<CT> frac(static_cast<size_t>(g.C * 3), CT(0));  // real-dim order
    for (int64_t c = 0; c < g.C; ++c)
        for (int d = 0; d < 3; ++d)
            if (g.wcol[d] >= 0)
                prep_forward<CT>(static_cast<CT>(w[c * g.nd + g.wcol[d]]), g.active, shifts[c * 3 + d], frac[c * 3 + g.wcol[d]]);
    const Maps maps = build_maps(g, g.S, shifts, -1);
    at::parallel_for(0, g.N * g.C, 1, [&](int64_t begin, int64_t end) {
        for (int64_t plane = begin; plane < end; ++plane) {
            const int64_t n = plane / g.C, c = plane % g.C;
            const int32_t *mp[3] = {maps.get(c, 0), maps.get(c, 1), maps.get(c, 2)};
            const T *xp = x + n * g.xs[0] + c * g.xs[1];
            T *op = out + n * g.os[0] + c * g.os[1];
            const CT *dw = frac.data() + c * 3;
            for (int64_t o0 = 0; o0 < g.O[0]; ++o0)
                for (int64_t o1 = 0; o1 < g.O[1]; ++o1)
                    for (int64_t o2 = 0; o2 < g.O[2]; ++o2) {
                        const int64_t p[3] = {o0 + g.L[0], o1 + g.L[1], o2 + g.L[2]};
                        T *o = op + o0 * g.os[2] + o1 * g.os[3] + o2 * g.os[4];
                        if (g.active) {
                            CT v[8];
                            corners<T, CT>(g.nd, xp, g.xs + 2, mp, p, true, v);
                            *o = static_cast<T>(interp_nd<CT>(g.nd, v, dw));
                        } else {
                            const int32_t a = mp[0][p[0]], b = mp[1][p[1]], e = mp[2][p[2]];
                            *o = (a >= 0 && b >= 0 && e >= 0) ? xp[a * g.xs[2] + b * g.xs[3] + e * g.xs[4]] : T(0);
                        }
                    }
        }
    });
}

template <typename T>
void backward_float(const Geom &g, const Tensor &grad, const Tensor &input, const Tensor &weights, Tensor &grad_input,
                    Tensor &grad_weights) {
    using CT = typename compute_of<T>::type;
    const T *go = grad.data_ptr<T>();
    const T *x = input.data_ptr<T>();
    T *gx = grad_input.data_ptr<T>();
    Tensor wc = weights.contiguous();
    const T *w = wc.data_ptr<T>();
    std::vector<int64_t> shifts(static_cast<size_t>(g.C * 3), 0);
    std::vector<CT> frac(static_cast<size_t>(g.C * 3), CT(0));
    for (int64_t c = 0; c < g.C; ++c)
        for (int d = 0; d < 3; ++d)
            if (g.wcol[d] >= 0)
                prep_backward<CT>(static_cast<CT>(w[c * g.nd + g.wcol[d]]), g.active, shifts[c * 3 + d], frac[c * 3 + g.wcol[d]]);
    const Maps xmaps = build_maps(g, g.S, shifts, -1);
    const Maps gmaps = build_maps(g, g.O, shifts, g.active ? -1 : +1);  // shifts_kernels.h:287-293
    std::vector<double> partial(static_cast<size_t>(g.N * g.C * 3), 0.0);
    at::parallel_for(0, g.N * g.C, 1, [&](int64_t begin, int64_t end) {
        for (int64_t plane = begin; plane < end; ++plane) {
            const int64_t n = plane / g.C, c = plane % g.C;
            const int32_t *xm[3] = {xmaps.get(c, 0), xmaps.get(c, 1), xmaps.get(c, 2)};
            const int32_t *gm[3] = {gmaps.get(c, 0), gmaps.get(c, 1), gmaps.get(c, 2)};
            const T *xp = x + n * g.xs[0] + c * g.xs[1];
            const T *gp = go + n * g.os[0] + c * g.os[1];
            T *gxp = gx + n * g.gs[0] + c * g.gs[1];
            const CT *dw = frac.data() + c * 3;
            double acc[3] = {0.0, 0.0, 0.0};
            for (int64_t i0 = 0; i0 < g.S[0]; ++i0)
                for (int64_t i1 = 0; i1 < g.S[1]; ++i1)
                    for (int64_t i2 = 0; i2 < g.S[2]; ++i2) {
                        const int64_t in[3] = {i0, i1, i2};
                        const int64_t o[3] = {i0 - g.L[0], i1 - g.L[1], i2 - g.L[2]};
                        const bool pass = o[0] >= 0 && o[0] < g.O[0] && o[1] >= 0 && o[1] < g.O[1] && o[2] >= 0 && o[2] < g.O[2];
                        T *dst = gxp + i0 * g.gs[2] + i1 * g.gs[3] + i2 * g.gs[4];
                        if (!pass) {
                            *dst = T(0);
                            continue;
                        }
                        const CT gval = static_cast<CT>(gp[o[0] * g.os[2] + o[1] * g.os[3] + o[2] * g.os[4]]);
                        CT v[8], wg[3];
                        corners<T, CT>(g.nd, xp, g.xs + 2, xm, in, true, v);
                        weight_grads_nd<CT>(g.nd, v, dw, wg);
                        for (int s = 0; s < g.nd; ++s) acc[s] += static_cast<double>(gval * wg[s]);
                        if (g.active) {
                            corners<T, CT>(g.nd, gp, g.os + 2, gm, o, true, v);
                            *dst = static_cast<T>(interp_nd<CT>(g.nd, v, dw));
                        } else {
                            const int32_t a = gm[0][o[0]], b = gm[1][o[1]], e = gm[2][o[2]];
                            *dst = (a >= 0 && b >= 0 && e >= 0) ? gp[a * g.os[2] + b * g.os[3] + e * g.os[4]] : T(0);
                        }
                    }
            for (int s = 0; s < 3; ++s) partial[plane * 3 + s] = acc[s];
        }
    });
    T *gw = grad_weights.data_ptr<T>();
    for (int64_t c = 0; c < g.C; ++c)
        for (int s = 0; s < g.nd; ++s) {
            double t = 0.0;
            for (int64_t n = 0; n < g.N; ++n) t += partial[(n * g.C + c) * 3 + s];
            gw[c * g.nd + s] = static_cast<T>(static_cast<CT>(t));
        }
}

template <typename R>
void forward_gather(const Geom &g, const R *x, const std::vector<int64_t> &shifts, R fill, R *out) {
    const Maps maps = build_maps(g, g.S, shifts, -1);
    at::parallel_for(0, g.N * g.C, 1, [&](int64_t begin, int64_t end) {
        for (int64_t plane = begin; plane < end; ++plane) {
            const int64_t n = plane / g.C, c = plane % g.C;
            const int32_t *m0 = maps.get(c, 0), *m1 = maps.get(c, 1), *m2 = maps.get(c, 2);
            const R *xp = x + n * g.xs[0] + c * g.xs[1];
            R *op = out + n * g.os[0] + c * g.os[1];
            for (int64_t o0 = 0; o0 < g.O[0]; ++o0)
                for (int64_t o1 = 0; o1 < g.O[1]; ++o1)
                    for (int64_t o2 = 0; o2 < g.O[2]; ++o2) {
                        const int32_t a = m0[o0 + g.L[0]], b = m1[o1 + g.L[1]], e = m2[o2 + g.L[2]];
                        op[o0 * g.os[2] + o1 * g.os[3] + o2 * g.os[4]] =
                            (a >= 0 && b >= 0 && e >= 0) ? xp[a * g.xs[2] + b * g.xs[3] + e * g.xs[4]] : fill;
                    }
        }
    });
}

void read_borders(const Tensor &borders, int32_t out[6]) {
    TORCH_CHECK(borders.numel() == 6, "borders must hold 6 integers [l_i, r_i, l_j, r_j, l_k, r_k]");
    Tensor b = borders.to(at::kCPU).to(at::kInt).contiguous();
    for (int i = 0; i < 6; ++i) out[i] = b.data_ptr<int32_t>()[i];
}

template <int ND> Tensor shift_forward_cpu(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                           at::IntArrayRef new_size, int64_t padding_mode, bool active_flag) {
    TORCH_CHECK(input.device().is_cpu() && weights.device().is_cpu(), "shiftnd_forward_cpu: expected CPU tensors");
    TORCH_CHECK(input.dim() == ND + 2, "shift", ND, "d: expected a ", ND + 2, "-D input");
    TORCH_CHECK(weights.dim() == 2 && weights.size(0) == input.size(1) && weights.size(1) == ND,
                "shift", ND, "d: weights must have shape [C, ", ND, "]");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();
    TORCH_CHECK(weights.scalar_type() == input.scalar_type(), "expected scalar type ", c10::toString(input.scalar_type()),
                " but found ", c10::toString(weights.scalar_type()));
    int32_t b[6];
    read_borders(borders, b);
    Tensor output = at::empty(new_size, input.options(), at::MemoryFormat::Contiguous);
    Geom g = make_geom(ND, input, b, padding_mode, active_flag);
    set_strides(output, ND, g.os);
    if (output.numel() == 0) return output;
    AT_DISPATCH_FLOATING_TYPES(input.scalar_type(), "shiftnd_forward_cpu",
                               [&] { forward_float<scalar_t>(g, input, weights, output); });
    return output;
}

template <int ND>
std::tuple<Tensor, Tensor> shift_backward_cpu(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                              const Tensor &borders, int64_t padding_mode, bool active_flag) {
    TORCH_CHECK(grad.device().is_cpu() && input.device().is_cpu() && weights.device().is_cpu(),
                "shiftnd_backward_cpu: expected CPU tensors");
    if (padding_mode < 0 || padding_mode > 4) return std::make_tuple(Tensor(), Tensor());
    TORCH_CHECK(weights.scalar_type() == input.scalar_type() && grad.scalar_type() == input.scalar_type(),
                "expected scalar type ", c10::toString(input.scalar_type()));
    int32_t b[6];
    read_borders(borders, b);
    Tensor grad_input = at::empty_like(input, at::MemoryFormat::Contiguous);
    Tensor grad_weights = at::zeros_like(weights, at::MemoryFormat::Contiguous);
    Geom g = make_geom(ND, input, b, padding_mode, active_flag);
    set_strides(grad, ND, g.os);
    set_strides(grad_input, ND, g.gs);
    if (input.numel() == 0) return std::make_tuple(grad_input, grad_weights);
    AT_DISPATCH_FLOATING_TYPES(grad.scalar_type(), "shiftnd_backward_cpu",
                               [&] { backward_float<scalar_t>(g, grad, input, weights, grad_input, grad_weights); });
    return std::make_tuple(grad_input, grad_weights);
}

// quantized/shifts_quantized.cpp:107-130
template <int ND> Tensor qshift_forward_cpu(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                            at::IntArrayRef new_size, int64_t padding_mode, bool /*active_flag*/) {
    TORCH_CHECK(input.is_quantized() && weights.is_quantized(), "q_shiftnd_cpu: expected quantized tensors");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();
    int32_t b[6];
    read_borders(borders, b);
    const int64_t wzp = weights.q_zero_point();
    Tensor iw = weights.int_repr().to(at::kLong).contiguous();
    TORCH_CHECK(iw.dim() == 2 && iw.size(0) == input.size(1) && iw.size(1) == ND, "shift", ND,
                "d: weights must have shape [C, ", ND, "]");
    const bool cl = input.is_contiguous(at::MemoryFormat::ChannelsLast) || input.is_contiguous(at::MemoryFormat::ChannelsLast3d);
    Tensor output = cl ? at::_empty_affine_quantized(new_size, input.options().memory_format(input.suggest_memory_format()),
                                                     input.q_scale(), input.q_zero_point(), c10::nullopt)
                       : at::_empty_affine_quantized(new_size, input.options(), input.q_scale(), input.q_zero_point());
    Geom g = make_geom(ND, input, b, padding_mode, false);
    set_strides(output, ND, g.os);
    if (output.numel() == 0) return output;
    std::vector<int64_t> shifts(static_cast<size_t>(g.C * 3), 0);
    for (int64_t c = 0; c < g.C; ++c)
        for (int d = 0; d < 3; ++d)
            if (g.wcol[d] >= 0) shifts[c * 3 + d] = iw.data_ptr<int64_t>()[c * ND + g.wcol[d]] - wzp;
    AT_DISPATCH_QINT_TYPES(input.scalar_type(), "q_shiftnd_cpu", [&] {
        const scalar_t fill = static_cast<scalar_t>(static_cast<underlying_t>(input.q_zero_point()));
        forward_gather<scalar_t>(g, input.data_ptr<scalar_t>(), shifts, fill, output.data_ptr<scalar_t>());
    });
    return output;
}

std::tuple<Tensor, Tensor> qshift_backward_cpu(const Tensor &, const Tensor &, const Tensor &, const Tensor &, int64_t, bool) {
    TORCH_CHECK(0, "backwards on quantized tensor are not supported");
}

}  // namespace cpu
}  // namespace torchshifts_amd

using namespace torchshifts_amd::cpu;

TORCH_LIBRARY_IMPL(torchshifts, CPU, m) {
    m.impl("_shift1d_forward", TORCH_FN(shift_forward_cpu<1>));
    m.impl("_shift1d_backward", TORCH_FN(shift_backward_cpu<1>));
    m.impl("_shift2d_forward", TORCH_FN(shift_forward_cpu<2>));
    m.impl("_shift2d_backward", TORCH_FN(shift_backward_cpu<2>));
    m.impl("_shift3d_forward", TORCH_FN(shift_forward_cpu<3>));
    m.impl("_shift3d_backward", TORCH_FN(shift_backward_cpu<3>));
}

TORCH_LIBRARY_IMPL(torchshifts, QuantizedCPU, m) {
    m.impl("_shift1d_forward", TORCH_FN(qshift_forward_cpu<1>));
    m.impl("_shift1d_backward", TORCH_FN(qshift_backward_cpu));
    m.impl("_shift2d_forward", TORCH_FN(qshift_forward_cpu<2>));
    m.impl("_shift2d_backward", TORCH_FN(qshift_backward_cpu));
    m.impl("_shift3d_forward", TORCH_FN(qshift_forward_cpu<3>));
    m.impl("_shift3d_backward", TORCH_FN(qshift_backward_cpu));
}
