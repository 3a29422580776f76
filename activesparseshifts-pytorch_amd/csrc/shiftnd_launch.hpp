// shiftnd_launch.hpp -- host-side entry points of the two kernel families (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shiftnd_common.hpp"

namespace shiftnd {

inline int dtype_size(int dtype) {
    switch (dtype) {
    case SHIFTND_F32: return 4;
    case SHIFTND_F64: return 8;
    case SHIFTND_F16: return 2;
    case SHIFTND_BF16: return 2;
    case SHIFTND_I8: return 1;
    case SHIFTND_U8: return 1;
    case SHIFTND_I32: return 4;
    default: return 0;
    }
}

// name of the main kernel launched by the last call on this host thread (diagnostics: bench.py matches it
// against the rocprofv3 kernel trace)
void note_kernel(const char *name);
const char *last_kernel();

// ---- strided fallback (shiftnd_strided.hip) ------------------------------------------------------
// wkind: dtype of the weights array (float dtypes -> rint / floor+frac; I8/U8/I32 -> repr - wzp).
int strided_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp,
                    uint64_t fill_bits, void *out, hipStream_t st);
size_t strided_backward_workspace(const Geometry &g);
int strided_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                     void *workspace, hipStream_t st);

// ---- per-plane kernels (shiftnd_plane.hip) -------------------------------------------------------
// *_eligible: contiguous NC[spatial] tensors, maps fit in LDS, planes < 2^31 elements, and (for the
// interpolating kernels) rows made of whole 16-byte chunks.
void plane_set_tuning(int knob, int value);
bool plane_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
bool plane_forward_lds_gather(const Geometry &g, int dtype, const void *x, const void *out);
int plane_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp,
                  uint64_t fill_bits, void *out, hipStream_t st);
bool plane_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t plane_backward_workspace(const Geometry &g, int dtype);
// 3-D volumes with rows that are not whole 16-byte pieces, beyond the small-plane kernels: the direct-load plane kernels with
// 4- / 8-byte chunks (interpolating forward, both backwards)
bool plane_ragged_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int plane_ragged_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st);
bool plane_ragged_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t plane_ragged_backward_workspace(const Geometry &g, int dtype);
int plane_ragged_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                          void *workspace, hipStream_t st);
int plane_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st);

// fused shift + average pool (kernel = stride = g.K, ceil mode), contiguous tensors only; the backward uses the
// plane_backward_workspace layout
bool plane_pool_forward_eligible(const Geometry &g, int dtype);
int plane_pool_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st);
bool plane_pool_backward_eligible(const Geometry &g, int dtype, const void *gx);
int plane_pool_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                        void *workspace, hipStream_t st);

// ---- sliding-window kernels (shiftnd_slide.hip): backward pass and interpolating forward of contiguous 2-D / 3-D
// problems without crop; part of the per-channel ("plane") family: plane_forward / plane_backward route to them
bool slide_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t slide_backward_workspace(const Geometry &g, int dtype);
int slide_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st);
bool slide_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int slide_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st);
void slide_set_tuning(int knob, int value);

// ---- one-step workgroups (shiftnd_step.hip): the backward pass of contiguous 2-D problems as a linear sweep of short
// workgroups; part of the per-channel family (plane_backward routes to it)
bool step_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
bool step_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t step_backward_workspace(const Geometry &g, int dtype);
int step_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                  void *workspace, hipStream_t st);
// ... and the sparse-shift / quantized forward of 4- / 8-byte elements in the same shape (part of the sweep family)
bool step_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int step_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                 void *out, hipStream_t st);
// ... and the forwards that stage their source rows in LDS: interpolating (every float dtype), sparse shift of 2-byte elements
bool step_forward_lds_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int step_forward_lds(const Geometry &g, int dtype, const void *x, const void *w, int wkind, uint64_t fill_bits, void *out, hipStream_t st);
void step_set_tuning(int knob, int value);
// ---- cropped 2-D windows and 1-D rows of any length as one-step workgroups over row spans (shiftnd_span.hip, round 4)
bool span_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int span_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
bool span_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t span_backward_workspace(const Geometry &g, int dtype);
bool span_forward_pooled3_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int span_forward_pooled3(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
bool span_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int span_forward_pooled(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
bool span_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t span_backward_pooled_workspace(const Geometry &g, int dtype);
int span_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                  hipStream_t st);
// ---- the 3-D backward of 16-bit tensors as a walk through the planes (shiftnd_walk.hip, round 4; plane_backward routes to it)
bool walk16_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int walk16_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
bool walk16_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t walk16_backward_workspace(const Geometry &g, int dtype);
int walk16_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                    hipStream_t st);

// ---- 1-byte elements on small planes (shiftnd_bytes.hip): whole planes through LDS, 16-byte output pieces that cross
// rows; part of the per-channel family (plane_forward routes to it)
bool bytes_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int bytes_forward(const Geometry &g, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                  hipStream_t st);
void bytes_set_tuning(int knob, int value);
// ... and one-byte planes that are NOT whole 16-byte pieces (14 x 14, 7 x 7): blocks of consecutive channels through LDS
bool bytes_block_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int bytes_block_forward(const Geometry &g, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                        hipStream_t st);

// ---- channel-fastest kernels for channels-last tensors (shiftnd_cl.hip): any strides, x with unit channel stride --
bool cl_forward_eligible(const Geometry &g);
bool cl_forward_preferred(const Geometry &g);   // eligible and the output is channel-fastest too
bool cl_backward_preferred(const Geometry &g, int dtype);
int cl_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
               void *out, hipStream_t st);
bool cl_backward_eligible(const Geometry &g, int dtype);
size_t cl_backward_workspace(const Geometry &g);
int cl_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                void *workspace, hipStream_t st);

// ---- LDS-tiled gather forward for dense channels-last inputs of 4-byte elements (shiftnd_cl_tiled.hip); the output is
// channels-last or NCHW-contiguous
bool cl_tiled_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int cl_tiled_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                     hipStream_t st);
void cl_tiled_set_tuning(int knob, int value);
// ... and the backward of 2-D fp32 problems whose three tensors are all dense channels-last
bool cl_tiled_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t cl_tiled_backward_workspace(const Geometry &g);
int cl_tiled_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                      void *workspace, hipStream_t st);

// ... and of 3-D problems whose saved input and grad_x are dense NDHWC, the gradient NDHWC or NCDHW-contiguous (shiftnd_cl_tiled3.hip, round 5)
bool cl_tiled3_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t cl_tiled3_backward_workspace(const Geometry &g);
int cl_tiled3_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                       void *workspace, hipStream_t st);

// ---- rows that are not whole 16-byte pieces as one-step workgroups over the tensor's flat chunk stream (shiftnd_flat.hip, round 5):
// contiguous 1-D / 2-D float tensors, a window included
bool flat_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int flat_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
bool flat_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t flat_backward_workspace(const Geometry &g);
int flat_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw, void *workspace,
                  hipStream_t st);
void flat_set_tuning(int value);   // knob 27: 0 automatic (ragged rows), 1 never, 2 whenever eligible

// ---- whole small planes through LDS (shiftnd_small.hip): interpolating forward and backward of contiguous problems whose
// rows are not whole 16-byte pieces (planes of at most 16 KiB)
bool small_forward_eligible(const Geometry &g, int dtype);
int small_forward(const Geometry &g, int dtype, const void *x, const void *w, void *out, hipStream_t st);
bool small_backward_eligible(const Geometry &g, int dtype);
size_t small_backward_workspace(const Geometry &g, int dtype);
int small_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st);
void small_set_tuning(int knob, int value);
// ... and the sparse-shift / quantized forward of 1-D / 2-D problems with such rows (any element size)
bool band_gather_forward_eligible(const Geometry &g, int dtype);
int band_gather_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                        void *out, hipStream_t st);

// ---- rows through LDS with register prefetch (shiftnd_rows.hip): sparse-shift / quantized forward of 1- / 2-byte elements,
// contiguous tensors, rows of whole 16-byte pieces
bool rows_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int rows_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits, void *out,
                 hipStream_t st);
void rows_set_tuning(int knob, int value);
bool bytes_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);

// ---- quantized shift + average pool in one pass (shiftnd_qpool.hip): one-byte element types, contiguous tensors
bool qpool_forward_eligible(const Geometry &g, int dtype);
bool walk_backward_pooled_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
bool walk_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
bool walk_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out);
bool step_forward_pooled_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int step_forward_pooled(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
int walk_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, void *out, hipStream_t st);
void qpool_set_tuning(int knob, int value);
int qpool_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, int64_t xzp, int requant, void *out,
                  hipStream_t st);

// ---- layout change (shiftnd_transpose.hip): dst[n][c][r] = src[n][r][c], dense tensors ---------------------------
int transpose_planes(const void *src, void *dst, int64_t N, int64_t rows, int64_t cols, int esize, hipStream_t st);

// ---- sweep kernels (shiftnd_sweep.hip): one 16-byte chunk per thread, XCD-contiguous grid ----------
bool sweep_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out);
int sweep_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                  void *out, hipStream_t st);
bool sweep_backward_eligible(const Geometry &g, int dtype, const void *go, const void *x, const void *gx);
size_t sweep_backward_workspace(const Geometry &g, int dtype);
int sweep_backward(const Geometry &g, int dtype, const void *go, const void *x, const void *w, void *gx, void *gw,
                   void *workspace, hipStream_t st);
void sweep_set_tuning(int knob, int value);
int sweep_debug_map(int64_t p, int64_t shift, int64_t len, int pad);

}  // namespace shiftnd
