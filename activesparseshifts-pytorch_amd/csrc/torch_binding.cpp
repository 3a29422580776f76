// torch_binding.cpp -- the `torchshifts` operator library for PyTorch-ROCm (host C++).
//
// Re-provides the reference's dispatcher surface so that `torchshifts/_C.so` is a drop-in:
//   * schemas of the six private ops + public shift{1,2,3}d + _cuda_version
//       (reference: csrc/ops/shifts.cpp:168-181, csrc/torchshifts.cpp:35-40)
//   * composite entry shift{N}d = check_borders + _shift{N}d_forward   (ops/shifts.cpp:93-166)
//   * Autograd key: one templated autograd::Function instead of three copies
//       (ops/autograd/shifts_autograd.cpp:15-281), incl. the double-backward guard
//   * CUDA key (= HIP tensors on PyTorch-ROCm) and QuantizedCUDA key: thin adapters that pull
//       pointers/strides out of at::Tensor and call the C ABI of libshiftnd_hip.so
//       (include/shiftnd_hip.h).  No compute happens here and there is no CPU fallback: a HIP
//       tensor either runs the HIP kernels or raises.
//   * New: shift{N}d_pool / _shift{N}d_pool_forward / _backward -- the shift followed by
//       the average pool the reference's modules attach (modules/shifts.py:81-89, 150-153) as ONE op.  On HIP
//       tensors it runs the fused kernels; everywhere else (CPU tensors, layouts the fused kernels do not
//       serve) it is literally the reference's two-step sequence.
// The CPU / QuantizedCPU keys (the reference's CPU backend) live in torch_cpu_backend.cpp.
#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_version.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <string>
#include <tuple>
#include <vector>

#include "shiftnd_hip.h"

namespace torchshifts_amd {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

using forward_sig = Tensor(const Tensor &, const Tensor &, const Tensor &, at::IntArrayRef, int64_t, bool);
using backward_sig = std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &, int64_t,
                                                bool);

// ---- dispatcher re-entry (ops/shifts.cpp:10-88) ------------------------------------------------------
template <int ND> Tensor call_forward(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                      at::IntArrayRef new_size, int64_t padding_mode, bool active_flag) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow(("torchshifts::_shift" + std::to_string(ND) + "d_forward").c_str(), "")
                         .typed<forward_sig>();
    return op.call(input, weights, borders, new_size, padding_mode, active_flag);
}
template <int ND>
std::tuple<Tensor, Tensor> call_backward(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                         const Tensor &borders, int64_t padding_mode, bool active_flag) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow(("torchshifts::_shift" + std::to_string(ND) + "d_backward").c_str(), "")
                         .typed<backward_sig>();
    return op.call(grad, weights, input, borders, padding_mode, active_flag);
}

// ---- borders -------------------------------------------------------------------------------------------
// check_borders (ops/shifts.cpp:93-135).  Unlike the reference the 6-int result stays on the HOST:
// it is host-known, and the kernels take it as launch arguments (no 24-byte H2D copy per call).
std::tuple<Tensor, std::vector<int64_t>> check_borders(const Tensor &input, const Tensor &borders, int64_t dim) {
    const auto sizes = input.sizes();
    TORCH_CHECK(static_cast<int64_t>(sizes.size()) >= dim + 1, "shift", dim, "d: input has too few dimensions");
    std::vector<int32_t> user;
    if (borders.numel() != 0) {
        Tensor b = borders.to(at::kInt).to(at::kCPU).contiguous();
        TORCH_CHECK(b.numel() >= 2 * std::min<int64_t>(dim, 3), "borders must hold (left, right) per spatial dim");
        user.assign(b.data_ptr<int32_t>(), b.data_ptr<int32_t>() + b.numel());
    }
    Tensor std_borders = at::empty({6}, at::TensorOptions().dtype(at::kInt).device(at::kCPU));
    std::vector<int64_t> new_sizes(sizes.size());
    const int rc = shiftnd_check_borders(sizes.data(), static_cast<int>(sizes.size()), user.empty() ? nullptr : user.data(),
                                         static_cast<int>(dim), std_borders.data_ptr<int32_t>(), new_sizes.data());
    TORCH_CHECK(rc == SHIFTND_OK, "check_borders: ", shiftnd_status_string(rc));
    new_sizes.resize(((dim + 1) == static_cast<int64_t>(sizes.size()) ? 1 : 2) + std::min<int64_t>(dim, 3));
    return std::make_tuple(std_borders, new_sizes);
}

// 6 host ints from a borders tensor that may live anywhere (a device tensor costs one D2H sync;
// the composite ops above never produce one)
void read_borders(const Tensor &borders, int32_t out[6]) {
    TORCH_CHECK(borders.numel() == 6, "borders must hold 6 integers [l_i, r_i, l_j, r_j, l_k, r_k]");
    Tensor b = borders.to(at::kCPU).to(at::kInt).contiguous();
    for (int i = 0; i < 6; ++i) out[i] = b.data_ptr<int32_t>()[i];
}

// The reference hands the private ops a 6-int DEVICE tensor that its kernels read (cuda/shifts_cuda.cu:61-67, ops/shifts.cpp:134).
// When the window has the input's own spatial sizes the borders are implied -- 0 <= l < r <= size and r - l == size leave
// l = 0, r = size -- so a reference-style caller stays free of the D2H sync and graph-capturable; a cropped window with
// device-resident borders still costs the one read (the kernels take the window as launch arguments).
void read_borders_for(const Tensor &borders, const Tensor &input, at::IntArrayRef window, int nd, int32_t out[6]) {
    TORCH_CHECK(borders.numel() == 6, "borders must hold 6 integers [l_i, r_i, l_j, r_j, l_k, r_k]");
    bool whole = !borders.is_cpu() && static_cast<int>(window.size()) == nd && input.dim() == nd + 2;
    for (int r = 0; whole && r < nd; ++r) whole = window[r] == input.size(2 + r);
    if (!whole) return read_borders(borders, out);
    for (int r = 0; r < 3; ++r) {
        out[2 * r] = 0;
        out[2 * r + 1] = r < nd ? static_cast<int32_t>(input.size(2 + r)) : 1;
    }
}

template <int ND> Tensor shift_public(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                      int64_t padding_mode, bool active_flag) {
    auto bands = check_borders(input, borders, ND);
    return call_forward<ND>(input, weights, std::get<0>(bands), std::get<1>(bands), padding_mode, active_flag);
}

std::tuple<Tensor, std::vector<int64_t>> check_borders_op(const Tensor &input, const Tensor &borders, int64_t dim) {
    return check_borders(input, borders, dim);
}

// ---- Autograd key (ops/autograd/shifts_autograd.cpp) -------------------------------------------------
bool is_channels_last_dense(const Tensor &t);
Tensor channels_last_to_contiguous(const Tensor &t);

template <int ND> struct ShiftFunction : public torch::autograd::Function<ShiftFunction<ND>> {
    static variable_list forward(AutogradContext *ctx, const Tensor &input, const Tensor &weight, const Tensor &borders,
                                 at::IntArrayRef new_size, int64_t padding_mode, bool active_flag) {
        at::AutoDispatchBelowADInplaceOrView guard;
        // (round 6) a dense NDHWC input on the GPU: the backward runs the contiguous kernels behind a layout change (the direct NDHWC
        // backward is slower), and the forward through the layout change costs what the direct forward costs
        // (N8 C128 16x112x112 fp32: 0.29 + 0.26 ms against 0.56; interpolating 0.57 against 0.71).  So the layout changes ONCE, here,
        // and the node keeps the contiguous copy instead of the input -- the backward's 0.29 ms transpose of x is gone (1.04 -> 0.73
        // ms with an NDHWC gradient, 0.74 -> 0.44 with the NCDHW gradient this forward's output produces); the values are the same,
        // the original input can be freed as soon as nobody else holds it.
        Tensor kept = input;
        if constexpr (ND == 3) {
            // (only when a backward can follow: an inference call keeps the direct NDHWC forward)
            if ((input.requires_grad() || weight.requires_grad()) && input.is_cuda() && !input.is_quantized() && input.dim() == 5 &&
                at::isFloatingType(input.scalar_type()) && is_channels_last_dense(input)) {
                c10::DeviceGuard device_guard(input.device());
                kept = channels_last_to_contiguous(input);
            }
        }
        auto output = call_forward<ND>(kept, weight, borders, new_size, padding_mode, active_flag);
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["active_flag"] = active_flag;
        ctx->save_for_backward({kept, weight, borders});
        return {output};
    }
    static variable_list backward(AutogradContext *ctx, const variable_list &grad_output) {
        auto saved = ctx->get_saved_variables();
        const auto padding_mode = ctx->saved_data["padding_mode"].toInt();
        const auto active_flag = ctx->saved_data["active_flag"].toBool();
        auto result = call_backward<ND>(grad_output[0], saved[1], saved[0], saved[2], padding_mode, active_flag);
        return {std::get<0>(result), std::get<1>(result), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// the backward op is itself differentiable only to the extent of raising on a second backward
template <int ND> struct ShiftBackwardFunction : public torch::autograd::Function<ShiftBackwardFunction<ND>> {
    static variable_list forward(AutogradContext *ctx, const Tensor &grad, const Tensor &weights, const Tensor &input,
                                 const Tensor &borders, int64_t padding_mode, bool active_flag) {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto result = call_backward<ND>(grad, weights, input, borders, padding_mode, active_flag);
        return {std::get<0>(result), std::get<1>(result)};
    }
    static variable_list backward(AutogradContext *, const variable_list &) {
        TORCH_CHECK(0, "double backwards on shift", ND, "d not supported");
    }
};

template <int ND> Tensor shift_autograd(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                        at::IntArrayRef new_size, int64_t padding_mode, bool active_flag) {
    return ShiftFunction<ND>::apply(input, weights, borders, new_size, padding_mode, active_flag)[0];
}
template <int ND>
std::tuple<Tensor, Tensor> shift_autograd_backward(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                                   const Tensor &borders, int64_t padding_mode, bool active_flag) {
    auto result = ShiftBackwardFunction<ND>::apply(grad, weights, input, borders, padding_mode, active_flag);
    return std::make_tuple(result[0], result[1]);
}

// ---- HIP backend adapters (CUDA dispatch key on PyTorch-ROCm) ----------------------------------------
int to_shiftnd_dtype(at::ScalarType t, const char *what) {
    switch (t) {
    case at::kFloat: return SHIFTND_F32;
    case at::kDouble: return SHIFTND_F64;
    case at::kHalf: return SHIFTND_F16;
    case at::kBFloat16: return SHIFTND_BF16;
    default: TORCH_CHECK(false, "\"", what, "\" not implemented for '", c10::toString(t), "'");
    }
    return -1;
}

void fill_strides(const Tensor &t, int nd, int64_t out[5]) {
    for (int i = 0; i < 5; ++i) out[i] = 0;
    for (int i = 0; i < 2 + nd; ++i) out[i] = t.stride(i);
}

void fill_problem(shiftnd_problem &p, int nd, const Tensor &input, const int32_t borders[6], int64_t padding_mode,
                  bool active, int dtype) {
    p.ndim = nd;
    p.dtype = dtype;
    p.padding_mode = static_cast<int32_t>(padding_mode);
    p.active = active ? 1 : 0;
    for (int i = 0; i < 5; ++i) p.sizes[i] = 1;
    for (int i = 0; i < 2 + nd; ++i) p.sizes[i] = input.size(i);
    for (int i = 0; i < 6; ++i) p.borders[i] = borders[i];
}

hipStream_t current_stream(const Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_same(const char *fn, const Tensor &a, const char *an, const Tensor &b, const char *bn) {
    TORCH_CHECK(a.get_device() == b.get_device(), "Expected tensor for ", an, " to be on the same device as tensor for ", bn,
                "; but ", an, " is on device ", a.get_device(), " and ", bn, " is on device ", b.get_device(),
                " (while checking arguments for ", fn, ")");
    TORCH_CHECK(a.scalar_type() == b.scalar_type(), "Expected tensor for ", an, " to have the same type as tensor for ", bn,
                "; but type ", c10::toString(a.scalar_type()), " does not equal ", c10::toString(b.scalar_type()),
                " (while checking arguments for ", fn, ")");
}

// ---- channels-last inputs ---------------------------------------------------------------------------------------
// The reference's CUDA backend walks a channels-last input through its strides (cuda/shifts_cuda.cu:217-262:
// uncoalesced) and returns an NCHW-contiguous result.  Here a dense channels-last tensor is first brought to the
// contiguous layout by shiftnd_transpose (a tile transpose at copy bandwidth, several times faster than ATen's layout
// copy), then the contiguous kernels run: for N16 C256 224x224 fp32 0.30 + 0.29 ms instead of 3.4 ms through strides.
bool is_channels_last_dense(const Tensor &t) {
    if (t.dim() < 3 || t.numel() == 0 || t.size(1) < 2 || t.is_contiguous()) return false;
    if (t.stride(1) != 1) return false;
    int64_t expect = t.size(1);
    for (int64_t d = t.dim() - 1; d >= 2; --d) {
        if (t.size(d) != 1 && t.stride(d) != expect) return false;
        expect *= t.size(d);
    }
    return t.size(0) == 1 || t.stride(0) == expect;
}

int64_t spatial_volume(const Tensor &t) {
    int64_t p = 1;
    for (int64_t d = 2; d < t.dim(); ++d) p *= t.size(d);
    return p;
}

// channels-last dense -> new contiguous tensor with the same values (works on int_repr bytes for quantized tensors)
Tensor channels_last_to_contiguous(const Tensor &t) {
    Tensor out = t.is_quantized()
                     ? at::_empty_affine_quantized(t.sizes(), t.options().memory_format(at::MemoryFormat::Contiguous), t.q_scale(),
                                                   t.q_zero_point(), c10::nullopt)
                     : at::empty(t.sizes(), t.options(), at::MemoryFormat::Contiguous);
    const int rc = shiftnd_transpose(t.data_ptr(), out.data_ptr(), t.size(0), spatial_volume(t), t.size(1),
                                     static_cast<int32_t>(t.element_size()), current_stream(t));
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_transpose (HIP): ", shiftnd_status_string(rc));
    return out;
}

// contiguous `src` -> the channels-last dense tensor `dst` (same sizes)
void contiguous_to_channels_last(const Tensor &src, Tensor &dst) {
    const int rc = shiftnd_transpose(src.data_ptr(), dst.data_ptr(), src.size(0), src.size(1), spatial_volume(src),
                                     static_cast<int32_t>(src.element_size()), current_stream(src));
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_transpose (HIP): ", shiftnd_status_string(rc));
}

template <int ND> Tensor shift_forward_hip(const Tensor &input_, const Tensor &weights, const Tensor &borders,
                                           at::IntArrayRef new_size, int64_t padding_mode, bool active_flag) {
    TORCH_CHECK(input_.is_cuda(), "input must be a CUDA tensor");
    TORCH_CHECK(weights.is_cuda(), "weights must be a CUDA tensor");
    check_same("shiftnd_forward_cuda", input_, "input", weights, "weights");
    TORCH_CHECK(input_.dim() == ND + 2, "shift", ND, "d: expected a ", ND + 2, "-D input");
    TORCH_CHECK(weights.dim() == 2 && weights.size(0) == input_.size(1) && weights.size(1) == ND,
                "shift", ND, "d: weights must have shape [C, ", ND, "]");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();  // the reference's switch has no default
    c10::DeviceGuard device_guard(input_.device());
    const int dtype = to_shiftnd_dtype(input_.scalar_type(), "shiftnd_forward_cuda");
    int32_t b[6];
    if (static_cast<int>(new_size.size()) == ND + 2) read_borders_for(borders, input_, new_size.slice(2), ND, b);
    else read_borders(borders, b);
    Tensor w = weights.contiguous();
    Tensor output = at::empty(new_size, input_.options(), at::MemoryFormat::Contiguous);
    shiftnd_problem p;
    fill_problem(p, ND, input_, b, padding_mode, active_flag, dtype);
    int64_t xs[5], os[5];
    fill_strides(output, ND, os);
    // a dense channels-last input: the library may have a kernel that reads it as it lies and writes the NCHW result
    // (shiftnd_cl_tiled.hip); otherwise one tile transpose, then the contiguous kernels
    bool direct = false;
    if (is_channels_last_dense(input_)) {
        fill_strides(input_, ND, xs);
        direct = shiftnd_forward_serves_channels_last(&p, input_.data_ptr(), xs, output.data_ptr(), os) != 0;
    }
    const Tensor input = (is_channels_last_dense(input_) && !direct) ? channels_last_to_contiguous(input_) : input_;
    fill_strides(input, ND, xs);
    const int rc = shiftnd_forward(&p, input.data_ptr(), xs, w.data_ptr(), output.data_ptr(), os, current_stream(input));
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_forward (HIP): ", shiftnd_status_string(rc));
    return output;
}

template <int ND>
std::tuple<Tensor, Tensor> shift_backward_hip(const Tensor &grad_, const Tensor &weights, const Tensor &input_,
                                              const Tensor &borders, int64_t padding_mode, bool active_flag) {
    TORCH_CHECK(grad_.is_cuda(), "grad must be a CUDA tensor");
    TORCH_CHECK(input_.is_cuda(), "input must be a CUDA tensor");
    TORCH_CHECK(weights.is_cuda(), "weights must be a CUDA tensor");
    check_same("shiftnd_backward_cuda", grad_, "grad", input_, "input");
    check_same("shiftnd_backward_cuda", grad_, "grad", weights, "weights");
    TORCH_CHECK(input_.dim() == ND + 2 && grad_.dim() == ND + 2, "shift", ND, "d backward: expected ", ND + 2, "-D tensors");
    if (padding_mode < 0 || padding_mode > 4) return std::make_tuple(Tensor(), Tensor());
    c10::DeviceGuard device_guard(grad_.device());
    int32_t b[6];
    read_borders_for(borders, input_, grad_.sizes().slice(2), ND, b);
    for (int r = 0; r < ND; ++r)
        TORCH_CHECK(grad_.size(2 + r) == b[2 * r + 1] - b[2 * r], "shift", ND, "d backward: grad does not match borders");
    const int dtype = to_shiftnd_dtype(grad_.scalar_type(), "shiftnd_backward_cuda");
    Tensor w = weights.contiguous();
    Tensor grad_weights = at::empty_like(w, at::MemoryFormat::Contiguous);
    // saved input dense channels-last, incoming gradient channels-last too or NCHW-contiguous (what follows the reference's
    // float forward, which returns NCHW for a channels-last input): the library may have a kernel for that layout
    // (shiftnd_cl_tiled.hip; grad_x comes out in the input's layout); otherwise change the layout once, then the
    // contiguous kernels
    bool direct = false;
    Tensor grad_input;
    if (is_channels_last_dense(input_) && (is_channels_last_dense(grad_) || grad_.is_contiguous())) {
        grad_input = at::empty_like(input_, input_.suggest_memory_format());
        shiftnd_problem pd;
        fill_problem(pd, ND, input_, b, padding_mode, active_flag, dtype);
        int64_t gd[5], xd[5], gxd[5];
        fill_strides(grad_, ND, gd);
        fill_strides(input_, ND, xd);
        fill_strides(grad_input, ND, gxd);
        direct = shiftnd_backward_serves_channels_last(&pd, grad_.data_ptr(), gd, input_.data_ptr(), xd, grad_input.data_ptr(), gxd) != 0;
    }
    const Tensor input = (is_channels_last_dense(input_) && !direct) ? channels_last_to_contiguous(input_) : input_;
    const Tensor grad = (is_channels_last_dense(grad_) && !direct) ? channels_last_to_contiguous(grad_) : grad_;
    if (!direct) grad_input = at::empty_like(input, at::MemoryFormat::Contiguous);
    shiftnd_problem p;
    fill_problem(p, ND, input, b, padding_mode, active_flag, dtype);
    const size_t ws_bytes = shiftnd_backward_workspace_bytes(&p);
    Tensor workspace = at::empty({static_cast<int64_t>(ws_bytes)}, input.options().dtype(at::kByte));
    int64_t gs[5], xs[5], gxs[5];
    fill_strides(grad, ND, gs);
    fill_strides(input, ND, xs);
    fill_strides(grad_input, ND, gxs);
    const int rc = shiftnd_backward(&p, grad.data_ptr(), gs, input.data_ptr(), xs, w.data_ptr(), grad_input.data_ptr(), gxs,
                                    grad_weights.data_ptr(), workspace.data_ptr(), ws_bytes, current_stream(grad));
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_backward (HIP): ", shiftnd_status_string(rc));
    return std::make_tuple(grad_input, grad_weights);
}

// ---- shift + average pool as one op ------------------------------------------------------------------------
using pool_forward_sig = Tensor(const Tensor &, const Tensor &, const Tensor &, at::IntArrayRef, at::IntArrayRef, int64_t, bool);
using pool_backward_sig = std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                     at::IntArrayRef, int64_t, bool);
template <int ND> Tensor call_pool_forward(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                           at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode,
                                           bool active_flag) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow(("torchshifts::_shift" + std::to_string(ND) + "d_pool_forward").c_str(), "")
                         .typed<pool_forward_sig>();
    return op.call(input, weights, borders, new_size, pool, padding_mode, active_flag);
}
template <int ND>
std::tuple<Tensor, Tensor> call_pool_backward(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                              const Tensor &borders, at::IntArrayRef pool, int64_t padding_mode,
                                              bool active_flag) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow(("torchshifts::_shift" + std::to_string(ND) + "d_pool_backward").c_str(), "")
                         .typed<pool_backward_sig>();
    return op.call(grad, weights, input, borders, pool, padding_mode, active_flag);
}

template <int ND> void check_pool(at::IntArrayRef pool) {
    TORCH_CHECK(static_cast<int>(pool.size()) == ND, "shift", ND, "d_pool: pool must hold ", ND, " window sizes");
    for (auto k : pool) TORCH_CHECK(k >= 1, "shift", ND, "d_pool: window sizes must be >= 1");
}

// the reference's sequence: _reduction_fn(shift(x)) with avg_pool{N}d(kernel = stride = pool, ceil_mode=True)
// (modules/shifts.py:81-89, 150-153)
template <int ND> Tensor pool_forward_composed(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                               at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode,
                                               bool active_flag) {
    check_pool<ND>(pool);
    Tensor y = call_forward<ND>(input, weights, borders, new_size, padding_mode, active_flag);
    if (!y.defined()) return y;
    if constexpr (ND == 1) return at::avg_pool1d(y, pool, pool, {0}, /*ceil_mode=*/true, /*count_include_pad=*/true);
    else if constexpr (ND == 2) return at::avg_pool2d(y, pool, pool, {0, 0}, true, true, c10::nullopt);
    else return at::avg_pool3d(y, pool, pool, {0, 0, 0}, true, true, c10::nullopt);
}

template <int ND>
std::tuple<Tensor, Tensor> pool_backward_composed(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                                  const Tensor &borders, at::IntArrayRef pool, int64_t padding_mode,
                                                  bool active_flag) {
    check_pool<ND>(pool);
    int32_t b[6];
    read_borders(borders, b);
    std::vector<int64_t> ysize = {input.size(0), input.size(1)};
    for (int r = 0; r < ND; ++r) ysize.push_back(b[2 * r + 1] - b[2 * r]);
    Tensor y_like = at::empty(ysize, grad.options());  // avg_pool's backward only looks at its shape
    Tensor g;
    if constexpr (ND == 1) {
        g = at::avg_pool2d_backward(grad.unsqueeze(2), y_like.unsqueeze(2), {1, pool[0]}, {1, pool[0]}, {0, 0}, true, true,
                                    c10::nullopt).squeeze(2);
    } else if constexpr (ND == 2) {
        g = at::avg_pool2d_backward(grad, y_like, pool, pool, {0, 0}, true, true, c10::nullopt);
    } else {
        g = at::avg_pool3d_backward(grad, y_like, pool, pool, {0, 0, 0}, true, true, c10::nullopt);
    }
    return call_backward<ND>(g, weights, input, borders, padding_mode, active_flag);
}

template <int ND> Tensor pool_forward_hip(const Tensor &input_, const Tensor &weights, const Tensor &borders,
                                          at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode,
                                          bool active_flag) {
    check_pool<ND>(pool);
    TORCH_CHECK(input_.is_cuda(), "input must be a CUDA tensor");
    TORCH_CHECK(weights.is_cuda(), "weights must be a CUDA tensor");
    check_same("shiftnd_pool_forward_cuda", input_, "input", weights, "weights");
    TORCH_CHECK(input_.dim() == ND + 2, "shift", ND, "d_pool: expected a ", ND + 2, "-D input");
    TORCH_CHECK(weights.dim() == 2 && weights.size(0) == input_.size(1) && weights.size(1) == ND,
                "shift", ND, "d_pool: weights must have shape [C, ", ND, "]");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();
    c10::DeviceGuard device_guard(input_.device());
    const Tensor input = is_channels_last_dense(input_) ? channels_last_to_contiguous(input_) : input_;
    if (!input.is_contiguous()) return pool_forward_composed<ND>(input, weights, borders, new_size, pool, padding_mode, active_flag);
    const int dtype = to_shiftnd_dtype(input.scalar_type(), "shiftnd_pool_forward_cuda");
    int32_t b[6], k[3] = {1, 1, 1};
    read_borders(borders, b);
    for (int r = 0; r < ND; ++r) k[r] = static_cast<int32_t>(pool[r]);
    shiftnd_problem p;
    fill_problem(p, ND, input, b, padding_mode, active_flag, dtype);
    int64_t ps[3];
    int rc = shiftnd_pooled_sizes(&p, k, ps);
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_pooled_sizes: ", shiftnd_status_string(rc));
    std::vector<int64_t> osize = {input.size(0), input.size(1)};
    for (int r = 0; r < ND; ++r) osize.push_back(ps[r]);
    Tensor w = weights.contiguous();
    Tensor output = at::empty(osize, input.options(), at::MemoryFormat::Contiguous);
    rc = shiftnd_forward_pooled(&p, k, input.data_ptr(), w.data_ptr(), output.data_ptr(), current_stream(input));
    if (rc == SHIFTND_ERR_NOT_FUSED)
        return pool_forward_composed<ND>(input, weights, borders, new_size, pool, padding_mode, active_flag);
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_forward_pooled (HIP): ", shiftnd_status_string(rc));
    return output;
}

template <int ND>
std::tuple<Tensor, Tensor> pool_backward_hip(const Tensor &grad, const Tensor &weights, const Tensor &input_,
                                             const Tensor &borders, at::IntArrayRef pool, int64_t padding_mode,
                                             bool active_flag) {
    check_pool<ND>(pool);
    TORCH_CHECK(grad.is_cuda(), "grad must be a CUDA tensor");
    TORCH_CHECK(input_.is_cuda(), "input must be a CUDA tensor");
    TORCH_CHECK(weights.is_cuda(), "weights must be a CUDA tensor");
    check_same("shiftnd_pool_backward_cuda", grad, "grad", input_, "input");
    check_same("shiftnd_pool_backward_cuda", grad, "grad", weights, "weights");
    TORCH_CHECK(input_.dim() == ND + 2 && grad.dim() == ND + 2, "shift", ND, "d_pool backward: expected ", ND + 2, "-D tensors");
    if (padding_mode < 0 || padding_mode > 4) return std::make_tuple(Tensor(), Tensor());
    c10::DeviceGuard device_guard(grad.device());
    const Tensor input = is_channels_last_dense(input_) ? channels_last_to_contiguous(input_) : input_;
    if (!input.is_contiguous())
        return pool_backward_composed<ND>(grad, weights, input, borders, pool, padding_mode, active_flag);
    const int dtype = to_shiftnd_dtype(grad.scalar_type(), "shiftnd_pool_backward_cuda");
    int32_t b[6], k[3] = {1, 1, 1};
    read_borders(borders, b);
    for (int r = 0; r < ND; ++r) k[r] = static_cast<int32_t>(pool[r]);
    shiftnd_problem p;
    fill_problem(p, ND, input, b, padding_mode, active_flag, dtype);
    int64_t ps[3];
    int rc = shiftnd_pooled_sizes(&p, k, ps);
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_pooled_sizes: ", shiftnd_status_string(rc));
    for (int r = 0; r < ND; ++r)
        TORCH_CHECK(grad.size(2 + r) == ps[r], "shift", ND, "d_pool backward: grad does not match the pooled size");
    Tensor g = grad.contiguous();
    Tensor w = weights.contiguous();
    Tensor grad_input = at::empty_like(input, at::MemoryFormat::Contiguous);
    Tensor grad_weights = at::empty_like(w, at::MemoryFormat::Contiguous);
    const size_t ws_bytes = shiftnd_backward_pooled_workspace_bytes(&p, k);  // (the pooled plan, not the plain one)
    Tensor workspace = at::empty({static_cast<int64_t>(ws_bytes)}, input.options().dtype(at::kByte));
    rc = shiftnd_backward_pooled(&p, k, g.data_ptr(), input.data_ptr(), w.data_ptr(), grad_input.data_ptr(),
                                 grad_weights.data_ptr(), workspace.data_ptr(), ws_bytes, current_stream(grad));
    if (rc == SHIFTND_ERR_NOT_FUSED)
        return pool_backward_composed<ND>(grad, weights, input, borders, pool, padding_mode, active_flag);
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_backward_pooled (HIP): ", shiftnd_status_string(rc));
    return std::make_tuple(grad_input, grad_weights);
}

template <int ND> struct ShiftPoolFunction : public torch::autograd::Function<ShiftPoolFunction<ND>> {
    static variable_list forward(AutogradContext *ctx, const Tensor &input, const Tensor &weight, const Tensor &borders,
                                 at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode, bool active_flag) {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto output = call_pool_forward<ND>(input, weight, borders, new_size, pool, padding_mode, active_flag);
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["active_flag"] = active_flag;
        ctx->saved_data["pool"] = pool.vec();
        ctx->save_for_backward({input, weight, borders});
        return {output};
    }
    static variable_list backward(AutogradContext *ctx, const variable_list &grad_output) {
        auto saved = ctx->get_saved_variables();
        const auto padding_mode = ctx->saved_data["padding_mode"].toInt();
        const auto active_flag = ctx->saved_data["active_flag"].toBool();
        const auto pool = ctx->saved_data["pool"].toIntVector();
        auto result = call_pool_backward<ND>(grad_output[0], saved[1], saved[0], saved[2], pool, padding_mode, active_flag);
        return {std::get<0>(result), std::get<1>(result), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
template <int ND> struct ShiftPoolBackwardFunction : public torch::autograd::Function<ShiftPoolBackwardFunction<ND>> {
    static variable_list forward(AutogradContext *ctx, const Tensor &grad, const Tensor &weights, const Tensor &input,
                                 const Tensor &borders, at::IntArrayRef pool, int64_t padding_mode, bool active_flag) {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto result = call_pool_backward<ND>(grad, weights, input, borders, pool, padding_mode, active_flag);
        return {std::get<0>(result), std::get<1>(result)};
    }
    static variable_list backward(AutogradContext *, const variable_list &) {
        TORCH_CHECK(0, "double backwards on shift", ND, "d not supported");
    }
};
template <int ND> Tensor pool_autograd(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                       at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode,
                                       bool active_flag) {
    return ShiftPoolFunction<ND>::apply(input, weights, borders, new_size, pool, padding_mode, active_flag)[0];
}
template <int ND>
std::tuple<Tensor, Tensor> pool_autograd_backward(const Tensor &grad, const Tensor &weights, const Tensor &input,
                                                  const Tensor &borders, at::IntArrayRef pool, int64_t padding_mode,
                                                  bool active_flag) {
    auto result = ShiftPoolBackwardFunction<ND>::apply(grad, weights, input, borders, pool, padding_mode, active_flag);
    return std::make_tuple(result[0], result[1]);
}
template <int ND> Tensor shift_pool_public(const Tensor &input, const Tensor &weights, const Tensor &borders,
                                           at::IntArrayRef pool, int64_t padding_mode, bool active_flag) {
    auto bands = check_borders(input, borders, ND);
    return call_pool_forward<ND>(input, weights, std::get<0>(bands), std::get<1>(bands), pool, padding_mode, active_flag);
}

// ---- quantized HIP forward (QuantizedCUDA key; new -- the reference only has QuantizedCPU) --------------
int quant_dtype(at::ScalarType t, const char *what) {
    switch (t) {
    case at::kQInt8: return SHIFTND_I8;
    case at::kQUInt8: return SHIFTND_U8;
    case at::kQInt32: return SHIFTND_I32;
    default: TORCH_CHECK(false, "\"", what, "\" not implemented for '", c10::toString(t), "'");
    }
    return -1;
}

template <int ND> Tensor qshift_forward_hip(const Tensor &input_, const Tensor &weights, const Tensor &borders,
                                            at::IntArrayRef new_size, int64_t padding_mode, bool /*active_flag*/) {
    TORCH_CHECK(input_.is_cuda() && input_.is_quantized(), "input must be a quantized CUDA tensor");
    TORCH_CHECK(weights.is_quantized(), "weights must be a quantized tensor");
    TORCH_CHECK(input_.dim() == ND + 2, "shift", ND, "d: expected a ", ND + 2, "-D input");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();
    c10::DeviceGuard device_guard(input_.device());
    const int dtype = quant_dtype(input_.scalar_type(), "q_shiftnd_cuda");
    const int wdtype = quant_dtype(weights.scalar_type(), "q_shiftnd_cuda");
    int32_t b[6];
    if (static_cast<int>(new_size.size()) == ND + 2) read_borders_for(borders, input_, new_size.slice(2), ND, b);
    else read_borders(borders, b);
    Tensor wrepr = weights.int_repr().to(input_.device()).contiguous();
    TORCH_CHECK(wrepr.dim() == 2 && wrepr.size(0) == input_.size(1) && wrepr.size(1) == ND,
                "shift", ND, "d: weights must have shape [C, ", ND, "]");
    const bool cl = input_.is_contiguous(at::MemoryFormat::ChannelsLast) || input_.is_contiguous(at::MemoryFormat::ChannelsLast3d);
    // a channels-last input keeps its format (shifts_quantized.cpp:119-121).  Two tile transposes around the contiguous
    // kernel beat the channel-fastest kernel (N128 C512 56x56 quint8: 0.09 + 0.14 + 0.09 ms against 0.69 ms)
    bool via_transpose = cl && is_channels_last_dense(input_);
    Tensor out_cl;  // the channels-last result: written by the LDS-tiled kernel, or by the layout change at the end
    if (via_transpose) {  // the LDS-tiled channels-last kernel keeps the format in one pass
        out_cl = at::_empty_affine_quantized(new_size, input_.options().memory_format(input_.suggest_memory_format()),
                                                    input_.q_scale(), input_.q_zero_point(), c10::nullopt);
        shiftnd_problem pd;
        fill_problem(pd, ND, input_, b, padding_mode, false, dtype);
        int64_t xd[5], od[5];
        fill_strides(input_, ND, xd);
        fill_strides(out_cl, ND, od);
        if (shiftnd_forward_serves_channels_last(&pd, input_.data_ptr(), xd, out_cl.data_ptr(), od)) {
            const int rcd = shiftnd_forward_quantized(&pd, input_.data_ptr(), xd, wrepr.data_ptr(), wdtype, weights.q_zero_point(),
                                                      input_.q_zero_point(), out_cl.data_ptr(), od, current_stream(input_));
            TORCH_CHECK(rcd == SHIFTND_OK, "shiftnd_forward_quantized (HIP): ", shiftnd_status_string(rcd));
            return out_cl;
        }
    }
    const Tensor input = via_transpose ? channels_last_to_contiguous(input_) : input_;
    Tensor output = (cl && !via_transpose)
                        ? at::_empty_affine_quantized(new_size, input.options().memory_format(input.suggest_memory_format()),
                                                      input.q_scale(), input.q_zero_point(), c10::nullopt)
                        : at::_empty_affine_quantized(new_size, input.options().memory_format(at::MemoryFormat::Contiguous),
                                                      input.q_scale(), input.q_zero_point(), c10::nullopt);
    shiftnd_problem p;
    fill_problem(p, ND, input, b, padding_mode, false, dtype);
    int64_t xs[5], os[5];
    fill_strides(input, ND, xs);
    fill_strides(output, ND, os);
    const int rc = shiftnd_forward_quantized(&p, input.data_ptr(), xs, wrepr.data_ptr(), wdtype, weights.q_zero_point(),
                                             input.q_zero_point(), output.data_ptr(), os, current_stream(input));
    TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_forward_quantized (HIP): ", shiftnd_status_string(rc));
    if (via_transpose && output.numel() > 0 && output.size(1) > 1) {
        contiguous_to_channels_last(output, out_cl);  // (allocated above: one channels-last buffer per call either way)
        return out_cl;
    }
    return output;
}

// quantized shift + average pool (the tail of a quantized module that emulates a strided depthwise conv) as one op on
// the QuantizedCUDA key: one pass for int8 / uint8 (shiftnd_forward_quantized_pooled); other element types and
// non-contiguous inputs run the quantized shift and then ATen's QuantizedCPU pool arithmetic with float HIP ops on the
// integer representation (the same values).
//
// ATen's QuantizedCPU average pool rounds in two ways (include/shiftnd_hip.h: SHIFTND_REQUANT_*): its channels-last kernel
// -- taken for every 3-D tensor and for a 4-D tensor that is_contiguous(ChannelsLast), which a contiguous one with C == 1
// or H == W == 1 also is; avg_pool1d pools the [N, C, 1, L] view -- adds the zero point after the rounding, its contiguous
// kernel before.  `y_sizes` / `y_channels_last`: the shift output the reference's module would hand to its pool
// (ops/quantized/shifts_quantized.cpp:119-125 keeps a channels-last input's layout).
template <int ND> bool qpool_zp_outside(at::IntArrayRef y_sizes, bool y_channels_last) {
    if (ND == 3) return true;
    if (ND == 2 && y_channels_last) return true;
    const int64_t C = y_sizes[1], H = ND == 1 ? 1 : y_sizes[2], W = y_sizes[ND + 1];
    return C == 1 || (H == 1 && W == 1);
}

template <int ND> Tensor qpool_composite(const Tensor &y, at::IntArrayRef pool) {
    const int64_t zp = y.q_zero_point();
    Tensor xi = y.int_repr();
    Tensor xf = xi.to(at::kFloat) - static_cast<double>(zp);
    std::vector<int64_t> k(pool.begin(), pool.end());
    if (ND == 1) {  // (avg_pool1d has no divisor_override: pool a [N, C, 1, L] view)
        xf = xf.unsqueeze(2);
        k.insert(k.begin(), 1);
    }
    std::vector<int64_t> zeros(k.size(), 0);
    // the window counts: ones of the SPATIAL shape (an empty batch or zero channels must not be indexed -- the reference's
    // sequence returns an empty tensor for them)
    std::vector<int64_t> one_shape(xf.sizes().begin(), xf.sizes().end());
    one_shape[0] = one_shape[1] = 1;
    Tensor ones = at::ones(one_shape, xf.options());
    Tensor sums = ND == 3 ? at::avg_pool3d(xf, k, k, zeros, true, true, 1) : at::avg_pool2d(xf, k, k, zeros, true, true, 1);
    Tensor cnt = ND == 3 ? at::avg_pool3d(ones, k, k, zeros, true, true, 1) : at::avg_pool2d(ones, k, k, zeros, true, true, 1);
    const bool outside = qpool_zp_outside<ND>(y.sizes(), ND == 2 && y.is_contiguous(at::MemoryFormat::ChannelsLast));
    Tensor mult = cnt.to(at::kDouble).reciprocal().to(at::kFloat);  // float(1 / count)
    Tensor res = outside ? at::round(sums * mult) + static_cast<double>(zp)
                         : at::round(sums * mult.reciprocal().reciprocal() + static_cast<double>(zp));
    if (ND == 1) res = res.squeeze(2);
    double lo = 0, hi = 255;
    if (xi.scalar_type() == at::kChar) { lo = -128; hi = 127; }
    if (xi.scalar_type() == at::kInt) { lo = -2147483648.0; hi = 2147483647.0; }
    res = res.clamp_(lo, hi).to(xi.scalar_type());
    return at::_make_per_tensor_quantized_tensor(res, y.q_scale(), zp);
}

template <int ND> Tensor qpool_forward_hip(const Tensor &input_, const Tensor &weights, const Tensor &borders,
                                           at::IntArrayRef new_size, at::IntArrayRef pool, int64_t padding_mode, bool active_flag) {
    check_pool<ND>(pool);
    TORCH_CHECK(input_.is_cuda() && input_.is_quantized(), "input must be a quantized CUDA tensor");
    TORCH_CHECK(weights.is_quantized(), "weights must be a quantized tensor");
    TORCH_CHECK(input_.dim() == ND + 2, "shift", ND, "d_pool: expected a ", ND + 2, "-D input");
    if (padding_mode < 0 || padding_mode > 4) return Tensor();
    c10::DeviceGuard device_guard(input_.device());
    const int dtype = quant_dtype(input_.scalar_type(), "q_shiftnd_pool_cuda");
    const int wdtype = quant_dtype(weights.scalar_type(), "q_shiftnd_pool_cuda");
    if (input_.is_contiguous() && (dtype == SHIFTND_I8 || dtype == SHIFTND_U8)) {
        int32_t b[6];
        read_borders(borders, b);
        Tensor wrepr = weights.int_repr().to(input_.device()).contiguous();
        TORCH_CHECK(wrepr.dim() == 2 && wrepr.size(0) == input_.size(1) && wrepr.size(1) == ND,
                    "shift", ND, "d_pool: weights must have shape [C, ", ND, "]");
        int32_t k[3] = {1, 1, 1};
        for (int r = 0; r < ND; ++r) k[r] = static_cast<int32_t>(pool[r]);
        shiftnd_problem p;
        fill_problem(p, ND, input_, b, padding_mode, false, dtype);
        int64_t ps[3];
        int rc = shiftnd_pooled_sizes(&p, k, ps);
        TORCH_CHECK(rc == SHIFTND_OK, "shiftnd_pooled_sizes: ", shiftnd_status_string(rc));
        std::vector<int64_t> osize = {input_.size(0), input_.size(1)};
        for (int r = 0; r < ND; ++r) osize.push_back(ps[r]);
        std::vector<int64_t> ysize = {input_.size(0), input_.size(1)};
        for (int r = 0; r < ND; ++r) ysize.push_back(b[2 * r + 1] - b[2 * r]);
        const int32_t requant = qpool_zp_outside<ND>(ysize, false) ? SHIFTND_REQUANT_ZP_OUTSIDE : SHIFTND_REQUANT_ZP_INSIDE;
        Tensor output = at::_empty_affine_quantized(osize, input_.options().memory_format(at::MemoryFormat::Contiguous),
                                                    input_.q_scale(), input_.q_zero_point(), c10::nullopt);
        rc = shiftnd_forward_quantized_pooled(&p, k, input_.data_ptr(), wrepr.data_ptr(), wdtype, weights.q_zero_point(),
                                              input_.q_zero_point(), requant, output.data_ptr(), current_stream(input_));
        if (rc == SHIFTND_OK) return output;
        TORCH_CHECK(rc == SHIFTND_ERR_NOT_FUSED, "shiftnd_forward_quantized_pooled (HIP): ", shiftnd_status_string(rc));
    }
    return qpool_composite<ND>(qshift_forward_hip<ND>(input_, weights, borders, new_size, padding_mode, active_flag), pool);
}

std::tuple<Tensor, Tensor> qshift_backward(const Tensor &, const Tensor &, const Tensor &, const Tensor &, int64_t, bool) {
    TORCH_CHECK(0, "backwards on quantized tensor are not supported");
}

int64_t cuda_version() { return -1; }  // no CUDA toolkit: extension.py only compares when torch.version.cuda is set
int64_t hip_version() { return HIP_VERSION; }

}  // namespace torchshifts_amd

using namespace torchshifts_amd;

TORCH_LIBRARY(torchshifts, m) {
    m.def("_cuda_version", &cuda_version);
    m.def("_hip_version", &hip_version);
    m.def("shift1d", &shift_public<1>);
    m.def("shift2d", &shift_public<2>);
    m.def("shift3d", &shift_public<3>);
    m.def("_check_borders", &check_borders_op);
    m.def("torchshifts::_shift1d_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift1d_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
    m.def("torchshifts::_shift2d_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift2d_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
    m.def("torchshifts::_shift3d_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift3d_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
    // shift + avg_pool(kernel = stride = pool, ceil_mode=True) as one op (not in the reference)
    m.def("torchshifts::shift1d_pool(Tensor input, Tensor weights, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> Tensor", &shift_pool_public<1>);
    m.def("torchshifts::shift2d_pool(Tensor input, Tensor weights, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> Tensor", &shift_pool_public<2>);
    m.def("torchshifts::shift3d_pool(Tensor input, Tensor weights, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> Tensor", &shift_pool_public<3>);
    m.def("torchshifts::_shift1d_pool_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int[] pool, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift1d_pool_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
    m.def("torchshifts::_shift2d_pool_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int[] pool, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift2d_pool_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
    m.def("torchshifts::_shift3d_pool_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, int[] pool, int padding_mode, bool active_flag) -> Tensor");
    m.def("torchshifts::_shift3d_pool_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, int[] pool, int padding_mode, bool active_flag) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(torchshifts, Autograd, m) {
    m.impl("_shift1d_forward", TORCH_FN(shift_autograd<1>));
    m.impl("_shift1d_backward", TORCH_FN(shift_autograd_backward<1>));
    m.impl("_shift2d_forward", TORCH_FN(shift_autograd<2>));
    m.impl("_shift2d_backward", TORCH_FN(shift_autograd_backward<2>));
    m.impl("_shift3d_forward", TORCH_FN(shift_autograd<3>));
    m.impl("_shift3d_backward", TORCH_FN(shift_autograd_backward<3>));
    m.impl("_shift1d_pool_forward", TORCH_FN(pool_autograd<1>));
    m.impl("_shift1d_pool_backward", TORCH_FN(pool_autograd_backward<1>));
    m.impl("_shift2d_pool_forward", TORCH_FN(pool_autograd<2>));
    m.impl("_shift2d_pool_backward", TORCH_FN(pool_autograd_backward<2>));
    m.impl("_shift3d_pool_forward", TORCH_FN(pool_autograd<3>));
    m.impl("_shift3d_pool_backward", TORCH_FN(pool_autograd_backward<3>));
}

// CPU tensors: the reference's two-step sequence (shift op on the CPU key, then ATen's pool)
TORCH_LIBRARY_IMPL(torchshifts, CPU, m) {
    m.impl("_shift1d_pool_forward", TORCH_FN(pool_forward_composed<1>));
    m.impl("_shift1d_pool_backward", TORCH_FN(pool_backward_composed<1>));
    m.impl("_shift2d_pool_forward", TORCH_FN(pool_forward_composed<2>));
    m.impl("_shift2d_pool_backward", TORCH_FN(pool_backward_composed<2>));
    m.impl("_shift3d_pool_forward", TORCH_FN(pool_forward_composed<3>));
    m.impl("_shift3d_pool_backward", TORCH_FN(pool_backward_composed<3>));
}

TORCH_LIBRARY_IMPL(torchshifts, CUDA, m) {
    m.impl("_shift1d_forward", TORCH_FN(shift_forward_hip<1>));
    m.impl("_shift1d_backward", TORCH_FN(shift_backward_hip<1>));
    m.impl("_shift2d_forward", TORCH_FN(shift_forward_hip<2>));
    m.impl("_shift2d_backward", TORCH_FN(shift_backward_hip<2>));
    m.impl("_shift3d_forward", TORCH_FN(shift_forward_hip<3>));
    m.impl("_shift3d_backward", TORCH_FN(shift_backward_hip<3>));
    m.impl("_shift1d_pool_forward", TORCH_FN(pool_forward_hip<1>));
    m.impl("_shift1d_pool_backward", TORCH_FN(pool_backward_hip<1>));
    m.impl("_shift2d_pool_forward", TORCH_FN(pool_forward_hip<2>));
    m.impl("_shift2d_pool_backward", TORCH_FN(pool_backward_hip<2>));
    m.impl("_shift3d_pool_forward", TORCH_FN(pool_forward_hip<3>));
    m.impl("_shift3d_pool_backward", TORCH_FN(pool_backward_hip<3>));
}

TORCH_LIBRARY_IMPL(torchshifts, QuantizedCUDA, m) {
    m.impl("_shift1d_forward", TORCH_FN(qshift_forward_hip<1>));
    m.impl("_shift1d_backward", TORCH_FN(qshift_backward));
    m.impl("_shift2d_forward", TORCH_FN(qshift_forward_hip<2>));
    m.impl("_shift2d_backward", TORCH_FN(qshift_backward));
    m.impl("_shift3d_forward", TORCH_FN(qshift_forward_hip<3>));
    m.impl("_shift3d_backward", TORCH_FN(qshift_backward));
    m.impl("_shift1d_pool_forward", TORCH_FN(qpool_forward_hip<1>));
    m.impl("_shift2d_pool_forward", TORCH_FN(qpool_forward_hip<2>));
    m.impl("_shift3d_pool_forward", TORCH_FN(qpool_forward_hip<3>));
}
