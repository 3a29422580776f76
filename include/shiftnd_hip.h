/*
 * shiftnd_hip.h -- C ABI of the MI355X-native shiftnd library (libshiftnd_hip.so).
 *
 * This is the drop-in boundary of the hot path: plain pointers, sizes and strides, no torch types.
 * Each entry point replaces one backend function of the reference
 * (DeadAt0m/ActiveSparseShifts-PyTorch, paths relative to torchshifts/csrc/ops/):
 *
 *   shiftnd_forward            <- shiftnd_forward<nD,pad,active>   cuda/shifts_cuda.cu:202-266
 *                                 (== cpu/shifts_cpu.cpp:216-232), i.e. the backend behind
 *                                 torchshifts::_shift{1,2,3}d_forward (shifts.cpp:168-181)
 *   shiftnd_backward           <- shiftnd_backward<nD,pad,active>  cuda/shifts_cuda.cu:270-345
 *                                 (== cpu/shifts_cpu.cpp:237-255), behind
 *                                 torchshifts::_shift{1,2,3}d_backward
 *   shiftnd_forward_quantized  <- qshiftnd<nD,pad>                 quantized/shifts_quantized.cpp:107-130
 *                                 (the reference has no GPU quantized path; this is new)
 *   shiftnd_check_borders      <- check_borders                    shifts.cpp:93-135 (host only)
 *   shiftnd_forward_pooled /   <- the module-level sequence `_reduction_fn(shift(x))` of the reference's
 *   shiftnd_backward_pooled       depthwise-conv emulation (torchshifts/modules/shifts.py:81-89, 150-153:
 *                                 shift, then avg_pool{N}d(kernel = stride, ceil_mode=True)) in one pass
 *
 * Conventions
 *   - Tensors are described the way the reference's kernels see them: sizes[5] = {N, C, H, W, D}
 *     of the INPUT (unused trailing spatial dims = 1) and element strides[5] in the same order
 *     (unused = 0).  ndim = number of spatial dims (1..3).
 *   - weights: device pointer to a contiguous [C, ndim] array of the input's dtype
 *     (column s shifts spatial dim s: H, W, D -- functional.py:76-77).
 *   - borders: 6 HOST ints {l_i, r_i, l_j, r_j, l_k, r_k}, the absolute [l, r) window that
 *     check_borders produces; the output has spatial sizes r - l.
 *   - padding_mode: 0 zeros, 1 border, 2 periodic, 3 reflect, 4 symmetric
 *     (BIPadding, kernels/shifts_kernels.h:5).
 *   - All device pointers must belong to the device that is current on the calling thread;
 *     `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue work: no host
 *     synchronisation, no allocation (graph-capture safe).
 *   - Every function returns SHIFTND_OK (0) or a negative shiftnd_status; nothing is thrown.
 *   - Numerics contract: SSL forward / SSL input-grad / quantized are pure
 *     gathers and bit-exact; for fp32 / fp64 tensors interpolation is evaluated as v1*(1-x)+v2*x
 *     with separate multiplies and add (no FMA contraction), bit-identical to the reference's CPU
 *     build; 16-bit inputs are widened to fp32, interpolated there (one multiply and one fused
 *     multiply-add per lerp) and rounded once (RNE) on store; the weight gradient sums products in
 *     the compute type (fused multiply-add), accumulates them in fp64 by a deterministic two-stage
 *     reduction and rounds once to the tensor dtype.
 */
#ifndef SHIFTND_HIP_H_
#define SHIFTND_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHIFTND_ABI_VERSION 5
#define SHIFTND_API __attribute__((visibility("default")))

typedef enum shiftnd_dtype {
    SHIFTND_F32 = 0,
    SHIFTND_F64 = 1,
    SHIFTND_F16 = 2,
    SHIFTND_BF16 = 3,
    SHIFTND_I8 = 4,  /* qint8  int_repr */
    SHIFTND_U8 = 5,  /* quint8 int_repr */
    SHIFTND_I32 = 6  /* qint32 int_repr */
} shiftnd_dtype;

typedef enum shiftnd_status {
    SHIFTND_OK = 0,
    SHIFTND_ERR_INVALID_ARGUMENT = -1,
    SHIFTND_ERR_UNSUPPORTED_DTYPE = -2,
    SHIFTND_ERR_WORKSPACE_TOO_SMALL = -3,
    SHIFTND_ERR_LAUNCH_FAILED = -4,
    SHIFTND_ERR_TOO_LARGE = -5,
    SHIFTND_ERR_NOT_FUSED = -6 /* pooled entry points: geometry not served; run shift and pool separately */
} shiftnd_status;

/* Which kernel family served the last call made on this host thread (for tests/benchmarks). */
typedef enum shiftnd_path {
    SHIFTND_PATH_NONE = 0,
    SHIFTND_PATH_EMPTY = 1,   /* zero-element problem: nothing launched */
    SHIFTND_PATH_PLANE = 2,   /* per-(N,C)-plane kernels, LDS index maps, 16-byte rows */
    SHIFTND_PATH_STRIDED = 3, /* generic strided fallback (channels-last, ragged rows, huge dims) */
    SHIFTND_PATH_SWEEP = 4,   /* one 16-byte chunk per thread, XCD-contiguous sweep (the HBM-rate path) */
    SHIFTND_PATH_CL = 5       /* channels-last input: lanes = consecutive channels of a pixel */
} shiftnd_path;

/* Problem geometry shared by the entry points. */
typedef struct shiftnd_problem {
    int32_t ndim;          /* spatial dims: 1, 2 or 3 */
    int32_t dtype;         /* shiftnd_dtype of input/output (and of float weights) */
    int32_t padding_mode;  /* 0..4 */
    int32_t active;        /* 0: sparse shift (rounded integer shift); 1: active (interpolated) */
    int64_t sizes[5];      /* input N, C, H, W, D */
    int32_t borders[6];    /* l_i, r_i, l_j, r_j, l_k, r_k (unused dims: 0, 1) */
} shiftnd_problem;

SHIFTND_API int shiftnd_abi_version(void);
SHIFTND_API const char *shiftnd_status_string(int status);
SHIFTND_API int shiftnd_last_path(void);
/* Diagnostics: name (without template arguments) of the main kernel the last call on this thread launched. */
SHIFTND_API const char *shiftnd_last_kernel(void);
/* 0 = automatic (sweep, else plane, else channels-last, else strided), 1 = force the strided fallback,
 * 2 = plane kernels or fail, 3 = sweep kernels or fail, 4 = channels-last kernels or fail (testing).
 * THREAD-LOCAL like shiftnd_last_path: the policy and the tuning knobs below apply to calls made on the calling
 * thread only, so a diagnostic setter can never re-route a concurrent caller (dispatcher threads, autograd
 * worker threads always run with the defaults). */
SHIFTND_API void shiftnd_set_path_policy(int policy);
/* Diagnostics: launch-planning knobs, by kernel family (csrc/shiftnd_api.hip routes them).  0-7 plane kernels
 * (0: minimum workgroups wanted, 1: target bytes per workgroup, 2: gather-forward unroll, 3: backward form, 5: affine
 * LDS reads, 6: XCD-contiguous block ids), 8-11 sweep kernels (9 / 11: max threads; the row steps per workgroup 8 / 10 once
 * chose are fixed), 12-15 sliding-window kernels (12: which problems take
 * them, 13: workgroups wanted, 14: minimum rows per band), 16-19 one-byte small-plane kernel (16: on / off, 17: planes
 * per round, 18: LDS bytes, 19: rounds per workgroup), 20-22 LDS-tiled channels-last kernels (20: on / off, 21: rows
 * per band, 22: XCD-contiguous block ids; 23: the direct NDHWC backward 0 = automatic, 1 = never, 2 = whenever eligible), 24-26 small-plane / row-band kernels (24: on / off, 25: planes per round or rows per band, 26: rounds per
 * workgroup), 27 flat-stream kernels for ragged rows (0 = automatic, 1 = never, 2 = whenever eligible), 28-30 one-byte row kernel (28: element sizes served, 29: rows per band, 30: workgroups wanted), 32-35
 * one-step kernels (32: 2-D backward, 33: sparse-shift forward by direct loads, 34: forwards through LDS; each 0 =
 * automatic, 1 = never, 2 = whenever eligible; 33 also: 3 = the 16-bit pooled gather forward off, 4 = the interpolating one off;
 * 35: bit set of opt-in forms, csrc/shiftnd_step.hip; round 6: bit 6 = the band-walk kernel for the cropped / interpolating 2-D pooled
 * backward, bit 7 = one row group per thread in crop_backward's sparse forms, bit 8 = two everywhere, bit 10 = the per-channel
 * kernels for cropped 3-D volumes and 2-D windows whose planes are not whole pieces, bit 11 = crop_backward3 / crop_forward3 instead of
 * the walk kernels with the window inside), 36-37 quantized
 * pool (36: 1 = the element-per-thread kernel only, 2 = the plane kernel first, 3 = the band kernel first; 37: workgroups wanted), 38 planes per workgroup of the 3-D walk
 * kernels; for the sizing knobs 0 means automatic.
 * Results never depend on them.
 * shiftnd_backward_workspace_bytes (and the pooled form) returns the larger of the default knobs' plan and the calling
 * thread's: a backward whose own plan needs more than it was given runs the default plan instead, so sizing on one
 * thread and running on another never ends in SHIFTND_ERR_WORKSPACE_TOO_SMALL. */
SHIFTND_API void shiftnd_set_tuning(int knob, int value);
/* Diagnostics: the sweep kernels' arithmetic padding map evaluated on the host: source index of
 * coordinate p (0 <= p <= len) under `shift`, or -1 for "fill". */
SHIFTND_API int shiftnd_debug_map(int64_t p, int64_t shift, int64_t len, int padding_mode);

/*
 * Host helper: the reference's check_borders (shifts.cpp:93-135).
 * sizes/nsizes: full tensor shape; user: ndim x 2 cut amounts (left, right) or NULL for no crop.
 * Writes borders[6] and new_sizes[nsizes].
 */
SHIFTND_API int shiftnd_check_borders(const int64_t *sizes, int nsizes, const int32_t *user, int ndim,
                          int32_t borders[6], int64_t *new_sizes);

/*
 * 1 when shiftnd_forward would serve this call with a kernel made for a dense channels-last INPUT (x_strides) and the
 * given output layout (channels-last or NCHW-contiguous) -- the caller then need not change the input's layout first
 * (the torch operator library transposes channels-last inputs otherwise); 0 for every other call.
 */
SHIFTND_API int shiftnd_forward_serves_channels_last(const shiftnd_problem *p, const void *x, const int64_t x_strides[5],
                                                     const void *out, const int64_t out_strides[5]);

/*
 * The same question for shiftnd_backward: 1 when the saved input, the incoming gradient and grad_x, all dense
 * channels-last as their strides say, are served by a kernel made for that layout (no layout change needed).
 */
SHIFTND_API int shiftnd_backward_serves_channels_last(const shiftnd_problem *p, const void *grad_out,
                                                      const int64_t grad_out_strides[5], const void *x,
                                                      const int64_t x_strides[5], const void *grad_x,
                                                      const int64_t grad_x_strides[5]);

/*
 * Forward, float dtypes (F32, F64, F16, BF16).
 * out has sizes {N, C, r_i-l_i, r_j-l_j, r_k-l_k}; out_strides are its element strides.
 */
SHIFTND_API int shiftnd_forward(const shiftnd_problem *p,
                    const void *x, const int64_t x_strides[5],
                    const void *weights,
                    void *out, const int64_t out_strides[5],
                    void *stream);

/*
 * Backward, float dtypes.  grad_out has the forward output's sizes; grad_x the input's sizes;
 * grad_w is a contiguous [C, ndim] array of the tensor dtype and is fully overwritten.
 * workspace: device scratch of at least shiftnd_backward_workspace_bytes(p) bytes
 * (fp64 partial sums of the weight gradient); its contents need not be initialised.
 */
SHIFTND_API size_t shiftnd_backward_workspace_bytes(const shiftnd_problem *p);

SHIFTND_API int shiftnd_backward(const shiftnd_problem *p,
                     const void *grad_out, const int64_t grad_out_strides[5],
                     const void *x, const int64_t x_strides[5],
                     const void *weights,
                     void *grad_x, const int64_t grad_x_strides[5],
                     void *grad_w,
                     void *workspace, size_t workspace_bytes,
                     void *stream);

/*
 * Quantized forward (dtype I8, U8 or I32 = int_repr of the quantized input; p->active ignored).
 * wq: device pointer to the contiguous [C, ndim] int_repr of the quantized weights, of type
 * wq_dtype (I8, U8 or I32); shift = wq - w_zero_point; scale is ignored
 * (kernels/shifts_kernels.h:553-555).  x_zero_point is the fill value (:569).
 */
SHIFTND_API int shiftnd_forward_quantized(const shiftnd_problem *p,
                              const void *x, const int64_t x_strides[5],
                              const void *wq, int32_t wq_dtype, int64_t w_zero_point,
                              int64_t x_zero_point,
                              void *out, const int64_t out_strides[5],
                              void *stream);

/*
 * Fused shift + average pool.  The reference's modules follow the shift with
 * avg_pool{N}d(kernel_size = stride = pool, ceil_mode=True) when they emulate a strided depthwise conv
 * (torchshifts/modules/shifts.py:81-89, 150-153); these entry points produce the same result without
 * writing / re-reading the full-size shift output.
 *   pool:   p->ndim ints, window (= stride) per spatial dim (H, W, D), each >= 1.
 *   Tensors are contiguous N, C, spatial: x and grad_x have the input's sizes, out / grad_pooled have spatial
 *   sizes ceil((r - l) / pool).  shiftnd_pooled_sizes writes them.
 *   Numerics: the window is summed in ATen's order in fp32 (fp64 for fp64) and divided by the number of window
 *   elements inside the shift output; fp32 / fp64 results are bit-identical to the two-step sequence.
 *   workspace: at least shiftnd_backward_pooled_workspace_bytes(p, pool) bytes (the pooled launch plan can need more
 *   partial-sum groups than the plain backward of the same tensor).
 * Return SHIFTND_ERR_NOT_FUSED when the geometry is not served (the caller then runs shift and pool
 * separately); nothing has been launched in that case.
 */
SHIFTND_API int shiftnd_pooled_sizes(const shiftnd_problem *p, const int32_t *pool, int64_t pooled_spatial[3]);

SHIFTND_API size_t shiftnd_backward_pooled_workspace_bytes(const shiftnd_problem *p, const int32_t *pool);

SHIFTND_API int shiftnd_forward_pooled(const shiftnd_problem *p, const int32_t *pool,
                           const void *x, const void *weights, void *out, void *stream);

SHIFTND_API int shiftnd_backward_pooled(const shiftnd_problem *p, const int32_t *pool,
                            const void *grad_pooled, const void *x, const void *weights,
                            void *grad_x, void *grad_w,
                            void *workspace, size_t workspace_bytes, void *stream);

#define SHIFTND_REQUANT_ZP_INSIDE 0
#define SHIFTND_REQUANT_ZP_OUTSIDE 1

/*
 * ABI 5: the same tail for the quantized forward (torchshifts/quantized/modules/shifts.py:19-20: a quantized module that
 * emulates a strided depthwise conv returns _reduction_fn(shift(x))): out = avg_pool(qshift(x)) in one pass, contiguous
 * int8 / uint8 tensors (p->dtype), quantized weights as in shiftnd_forward_quantized.  Requantization is ATen's
 * QuantizedCPU average pool in fp32 on acc = sum(x_int - x_zero_point) over the window, clamped to the type's range; scale
 * and zero point of the result are the input's.  ATen itself rounds in two ways and `requant` names the one wanted:
 *   SHIFTND_REQUANT_ZP_INSIDE   nearbyint(x_zero_point + acc * (1 / count))  -- its kernel for contiguous 1-D / 2-D tensors
 *                               (quantize_val -> fbgemm::Quantize: the zero point is added before the rounding)
 *   SHIFTND_REQUANT_ZP_OUTSIDE  nearbyint(acc * (1 / count)) + x_zero_point  -- its channels-last kernel, which also serves
 *                               every 3-D tensor and any [N, C, H, W] tensor whose shift output is channels-last-contiguous
 *                               too (C == 1 or H == W == 1)
 * (they differ when the zero point is odd and the window mean is a tie).  The torch operator library picks the form ATen
 * would use for the reference's shift output (torch_binding.cpp: qpool_zp_outside).  SHIFTND_ERR_NOT_FUSED for other element types (nothing launched).
 */
SHIFTND_API int shiftnd_forward_quantized_pooled(const shiftnd_problem *p, const int32_t *pool, const void *x, const void *wq,
                                     int32_t wq_dtype, int64_t w_zero_point, int64_t x_zero_point, int32_t requant, void *out,
                                     void *stream);

/*
 * Layout change between channels-last and contiguous tensors: dst[b][c][r] = src[b][r][c] for dense
 * src[batch][rows][cols] and dst[batch][cols][rows] of element_bytes-byte (1, 2, 4, 8) elements.
 * Channels-last [N, H, W, C] -> contiguous [N, C, H, W]: rows = H*W, cols = C; the reverse: rows = C, cols = H*W.
 * The reference's float ops return an NCHW-contiguous tensor even for a channels-last input
 * (cpu/shifts_cpu.cpp:221) and its CUDA backend walks such an input through its strides; host glue that meets a
 * channels-last tensor changes the layout with this call (a tile transpose at copy bandwidth) and runs the
 * contiguous kernels, which is several times faster than any kernel that touches both layouts at once.
 */
SHIFTND_API int shiftnd_transpose(const void *src, void *dst, int64_t batch, int64_t rows, int64_t cols,
                      int32_t element_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHIFTND_HIP_H_ */
