// shiftnd_api.hip -- the C ABI declared in include/shiftnd_hip.h: argument validation, geometry
// normalisation and kernel-family selection.  No torch types, no allocation, no synchronisation.
#include <string.h>

#include <algorithm>

#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

using namespace shiftnd;

namespace {

thread_local int g_last_path = SHIFTND_PATH_NONE;
thread_local const char *g_last_kernel = "";
thread_local int g_cl3 = 0;     // knob 23: the direct NDHWC backward (shiftnd_cl_tiled3.hip) 0 automatic / 1 never / 2 whenever eligible
thread_local int g_policy = 0;  // 0 auto, 1 force strided, 2 plane kernels (or fail), 3 sweep kernels (or fail)

bool is_float_dtype(int dt) { return dt >= SHIFTND_F32 && dt <= SHIFTND_BF16; }
bool is_quant_dtype(int dt) { return dt >= SHIFTND_I8 && dt <= SHIFTND_I32; }

// Build the normalised 3-dim geometry: real spatial dim r (0..nd-1 = H, W, D) becomes normalised
// dim r + 3 - nd, so the innermost (contiguous) dim is always index 2 and leading dims have size 1.
int build_geometry(const shiftnd_problem *p, const int64_t *xs, const int64_t *os, const int64_t *gs, Geometry &g) {
    if (!p || p->ndim < 1 || p->ndim > 3 || p->padding_mode < 0 || p->padding_mode > 4) return SHIFTND_ERR_INVALID_ARGUMENT;
    memset(&g, 0, sizeof(g));
    g.nd = p->ndim;
    g.pad = p->padding_mode;
    g.active = p->active ? 1 : 0;
    g.N = p->sizes[0];
    g.C = p->sizes[1];
    if (g.N < 0 || g.C < 0) return SHIFTND_ERR_INVALID_ARGUMENT;
    const int lead = 3 - p->ndim;
    for (int d = 0; d < 3; ++d) {
        g.S[d] = 1;
        g.O[d] = 1;
        g.L[d] = 0;
        g.wcol[d] = -1;
    }
    g.xs[0] = xs[0];
    g.xs[1] = xs[1];
    g.os[0] = os[0];
    g.os[1] = os[1];
    if (gs) {
        g.gs[0] = gs[0];
        g.gs[1] = gs[1];
    }
    for (int r = 0; r < p->ndim; ++r) {
        const int d = r + lead;
        const int64_t size = p->sizes[2 + r];
        const int64_t l = p->borders[2 * r], rr = p->borders[2 * r + 1];
        if (size < 0) return SHIFTND_ERR_INVALID_ARGUMENT;
        if (size > 0 && (l < 0 || rr > size || rr <= l)) return SHIFTND_ERR_INVALID_ARGUMENT;
        g.S[d] = size;
        g.L[d] = l;
        g.O[d] = size > 0 ? rr - l : 0;
        g.wcol[d] = r;
        g.xs[2 + d] = xs[2 + r];
        g.os[2 + d] = os[2 + r];
        if (gs) g.gs[2 + d] = gs[2 + r];
    }
    return SHIFTND_OK;
}

thread_local int g_flat = 0;   // knob 27 (mirrors shiftnd_flat.hip's): 2 = the flat-stream kernels whenever eligible

// rows of the INPUT that are not whole 16-byte pieces: what the chunk kernels (step_*, crop_*, row_*: input rows of whole pieces, any
// window) do not take -- the flat-stream kernels' automatic share
bool ragged_rows(const Geometry &g, int dtype) {
    const int es = dtype_size(dtype);
    return (g.S[2] * es) % 16 != 0;
}

bool cropped(const Geometry &g) {
    for (int d = 0; d < 3; ++d)
        if (g.L[d] != 0 || g.O[d] != g.S[d]) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
// The tuning knobs (shiftnd_set_tuning) live in thread-local arrays next to the kernels they shape; this file keeps a
// thread-local shadow of every value so that a call can be planned under the DEFAULT knobs whatever the calling thread
// has set: shiftnd_backward_workspace_bytes returns the larger of the default plan's and the calling thread's plan's
// bytes, and a backward whose plan (under the calling thread's knobs) needs more than it was given runs under the
// default plan instead -- sizing on one thread and running on another can never end in WORKSPACE_TOO_SMALL.
// ---------------------------------------------------------------------------------------------
constexpr int kKnobs = 40;
constexpr int kKnobDefault[kKnobs] = {
    0, 128 * 1024, 4, 2, 1, 0, 1, 0,   //  0..7   plane kernels (g_tune)
    4, 512, 2, 256,                    //  8..11  sweep kernels
    -1, 0, 16, 0,                      // 12..15  sliding-window kernels
    1, 0, 0, 0,                        // 16..19  one-byte small-plane kernel
    1, 0, 1,                           // 20..22  LDS-tiled channels-last kernels
    0,                                 // 23      direct NDHWC backward
    1, 0, 0,                           // 24..26  small-plane / row-band kernels
    0,                                 // 27      flat-stream kernels
    1, 0, 0, 0,                        // 28..31  one-byte row kernel (31 unused)
    0, 0, 0, 0,                        // 32..35  one-step kernels
    0, 0,                              // 36..37  quantized pool
    0, 0};                             // 38      planes per workgroup of the walk kernels (39 unused)
thread_local int g_knob[kKnobs] = {
    0, 128 * 1024, 4, 2, 1, 0, 1, 0, 4, 512, 2, 256, -1, 0, 16, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

void apply_knob(int knob, int value) {
    if (knob == 27) { g_flat = value; flat_set_tuning(value); }   // 27: the flat-stream kernels 0 automatic (ragged rows) / 1 never / 2 whenever eligible
    else if (knob >= 38) step_set_tuning(4 + knob - 38, value);  // 38: planes per workgroup of the walk kernels
    else if (knob >= 36) qpool_set_tuning(knob - 36, value);  // 36: quantized pool 0 automatic / 1 the element-per-thread kernel only
    else if (knob >= 32) step_set_tuning(knob - 32, value);  // 32: one-step backward 0 automatic / 1 never / 2 whenever eligible
    else if (knob >= 28) rows_set_tuning(knob - 28, value);  // 28: element sizes served (bit 0: 1 byte, bit 1: 2 bytes), 29: rows per band, 30: workgroups
    else if (knob >= 24) small_set_tuning(knob - 24, value);  // 24: small-plane kernels on / off, 25: planes per round, 26: rounds per workgroup
    else if (knob == 23) g_cl3 = value;
    else if (knob >= 20) cl_tiled_set_tuning(knob - 20, value);  // 20: LDS-tiled channels-last forward on / off, 21: rows per band
    else if (knob >= 16) bytes_set_tuning(knob - 16, value);  // 16: 1-byte small-plane kernel on / off, 17: planes per workgroup
    else if (knob >= 12) slide_set_tuning(knob - 12, value);  // 12: which problems slide, 13: workgroups wanted, 14: min rows per band
    else if (knob >= 8) sweep_set_tuning(knob - 8, value);  // 8/9: sweep forward K / max threads, 10/11: sweep backward
    else plane_set_tuning(knob, value);
}

bool knobs_touched() {
    for (int k = 0; k < kKnobs; ++k)
        if (g_knob[k] != kKnobDefault[k]) return true;
    return false;
}

// the default knobs for the lifetime of the object (this thread only), the thread's own back afterwards
struct DefaultKnobs {
    DefaultKnobs() {
        for (int k = 0; k < kKnobs; ++k)
            if (g_knob[k] != kKnobDefault[k]) apply_knob(k, kKnobDefault[k]);
    }
    ~DefaultKnobs() {
        for (int k = 0; k < kKnobs; ++k)
            if (g_knob[k] != kKnobDefault[k]) apply_knob(k, g_knob[k]);
    }
    DefaultKnobs(const DefaultKnobs &) = delete;
    DefaultKnobs &operator=(const DefaultKnobs &) = delete;
};

bool empty_problem(const Geometry &g) {
    return g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0 || g.O[0] * g.O[1] * g.O[2] == 0;
}

int finish(int rc) {
    if (rc != SHIFTND_OK) return rc;
    return hipGetLastError() == hipSuccess ? SHIFTND_OK : SHIFTND_ERR_LAUNCH_FAILED;
}

int forward_common(const shiftnd_problem *p, const void *x, const int64_t *xs, const void *w, int wkind, int64_t wzp,
                   uint64_t fill, void *out, const int64_t *os, void *stream) {
    Geometry g;
    const int rc = build_geometry(p, xs, os, nullptr, g);
    if (rc != SHIFTND_OK) return rc;
    if (empty_problem(g)) {
        g_last_path = SHIFTND_PATH_EMPTY;
        return SHIFTND_OK;
    }
    if (!x || !w || !out) return SHIFTND_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool can_sweep = sweep_forward_eligible(g, p->dtype, x, out);
    const bool can_plane = plane_forward_eligible(g, p->dtype, x, out);
    if ((g_policy == 2 && !can_plane) || (g_policy == 3 && !can_sweep)) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (g_policy != 4) {
        // automatic choice: the sweep kernels for planes of >= 32 KiB (their per-wave prologue is amortised over
        // several rows); small planes (e.g. 56x56 int8 = 3 KiB) go to the plane kernels, which walk many planes
        // of one channel per workgroup (measured on C4: 0.13 ms vs 0.33 ms)
        const int64_t out_plane_bytes = g.O[0] * g.O[1] * g.O[2] * dtype_size(p->dtype);
        // (interpolating problems keep the LDS-staged plane kernels whenever those take them: see DESIGN 3.14)
        const bool interpolating = g.active && p->dtype <= SHIFTND_BF16;
        const bool prefer_sweep = (out_plane_bytes >= 32 * 1024 && !interpolating && !(can_plane && plane_forward_lds_gather(g, p->dtype, x, out))) || !can_plane;
        if (g_policy == 0 && g_flat == 2 && wkind == p->dtype && flat_forward_eligible(g, p->dtype, x, out)) {   // (knob 27 = 2: tests)
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(flat_forward(g, p->dtype, x, w, wkind, out, st));
        }
        // 2-D sparse shift / quantized forward as a linear sweep of one-step workgroups
        if (g_policy == 0 && (g.nd == 2 || wkind <= SHIFTND_BF16) && step_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_SWEEP;
            return finish(step_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
        }
        // cropped windows with ragged rows, 1-D rows of any length, ragged source rows (float tensors): one-step workgroups over
        // row spans -- what the aligned one-step forwards below do not take
        // cropped 16-bit volumes under zeros padding, interpolating: the walk through the planes with the window inside (round 6)
        if (g_policy == 0 && wkind == p->dtype && cropped(g) && walk16_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(walk16_forward(g, p->dtype, x, w, wkind, out, st));
        }
        if (g_policy == 0 && wkind == p->dtype && !step_forward_lds_eligible(g, p->dtype, x, out) && span_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(span_forward(g, p->dtype, x, w, wkind, out, st));
        }
        // rows that are not whole 16-byte pieces (14 x 14, 62 x 62, 222 x 222 ...; float tensors): one-step workgroups over the
        // tensor's flat chunk stream
        if (g_policy == 0 && wkind == p->dtype && ragged_rows(g, p->dtype) && flat_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(flat_forward(g, p->dtype, x, w, wkind, out, st));
        }
        // 1-byte (and, knob 28, 2-byte) rows of whole 16-byte pieces beyond the byte kernel's small planes: rows through LDS
        if (g_policy == 0 && !bytes_forward_eligible(g, p->dtype, x, out) && rows_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(rows_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
        }
        // rows that are not whole 16-byte pieces, planes above 16 KiB: the chunk kernels would move them element by
        // element (shiftnd_small.hip)
        if (g_policy == 0 && out_plane_bytes > 16 * 1024 && band_gather_forward_eligible(g, p->dtype)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(band_gather_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
        }
        // 2-D sparse shift of 4- / 8-byte elements on planes of >= 32 KiB: the linear sweep of one-step workgroups
        // 3-D interpolating forward: a walk through the planes (one new plane per step, the other carried in registers)
        if (g_policy == 0 && wkind == p->dtype && walk16_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(walk16_forward(g, p->dtype, x, w, wkind, out, st));
        }
        if (g_policy == 0 && wkind <= SHIFTND_BF16 && walk_forward_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(walk_forward(g, p->dtype, x, w, wkind, out, st));
        }
        if (g_policy == 0 && wkind <= SHIFTND_BF16 && step_forward_lds_eligible(g, p->dtype, x, out)) {
            g_last_path = SHIFTND_PATH_PLANE;  // (LDS-staged, like the per-channel kernels it supersedes)
            return finish(step_forward_lds(g, p->dtype, x, w, wkind, fill, out, st));
        }
        if (can_sweep && (g_policy == 3 || (g_policy == 0 && prefer_sweep))) {
            g_last_path = SHIFTND_PATH_SWEEP;
            return finish(sweep_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
        }
        if (can_plane && g_policy != 1) {
            g_last_path = SHIFTND_PATH_PLANE;
            return finish(plane_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
        }
    }
    // windows no chunk kernel took (output planes that are not a whole number of 16-byte pieces): the flat chunk stream before the
    // one-thread-per-element kernels
    if (g_policy == 0 && wkind == p->dtype && cropped(g) && flat_forward_eligible(g, p->dtype, x, out)) {
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(flat_forward(g, p->dtype, x, w, wkind, out, st));
    }
    if (g_policy == 0 && small_forward_eligible(g, p->dtype)) {  // interpolating, rows not whole 16-byte pieces, small planes
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(small_forward(g, p->dtype, x, w, out, st));
    }
    if (g_policy == 0 && wkind == p->dtype && plane_ragged_forward_eligible(g, p->dtype, x, out)) {  // ... larger 3-D volumes: narrow chunks
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(plane_ragged_forward(g, p->dtype, x, w, out, st));
    }
    if ((g_policy == 0 || g_policy == 4) && cl_tiled_forward_eligible(g, p->dtype, x, out)) {  // channels-last in, LDS-tiled
        g_last_path = SHIFTND_PATH_CL;
        return finish(cl_tiled_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
    }
    if (g_policy == 4 && !cl_forward_eligible(g)) return SHIFTND_ERR_INVALID_ARGUMENT;
    if ((g_policy == 0 && cl_forward_preferred(g)) || g_policy == 4) {  // channels-last tensors: channel-fastest kernels
        g_last_path = SHIFTND_PATH_CL;
        return finish(cl_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
    }
    g_last_path = SHIFTND_PATH_STRIDED;
    return finish(strided_forward(g, p->dtype, x, w, wkind, wzp, fill, out, st));
}

}  // namespace

namespace shiftnd {
void note_kernel(const char *name) { g_last_kernel = name; }
const char *last_kernel() { return g_last_kernel; }
}  // namespace shiftnd

extern "C" {

const char *shiftnd_last_kernel(void) { return g_last_kernel; }

int shiftnd_abi_version(void) { return SHIFTND_ABI_VERSION; }

const char *shiftnd_status_string(int status) {
    switch (status) {
    case SHIFTND_OK: return "ok";
    case SHIFTND_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SHIFTND_ERR_UNSUPPORTED_DTYPE: return "unsupported dtype";
    case SHIFTND_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case SHIFTND_ERR_LAUNCH_FAILED: return "kernel launch failed";
    case SHIFTND_ERR_TOO_LARGE: return "problem too large";
    case SHIFTND_ERR_NOT_FUSED: return "geometry not served by the fused kernels";
    default: return "unknown status";
    }
}

int shiftnd_last_path(void) { return g_last_path; }

void shiftnd_set_path_policy(int policy) { g_policy = policy; }

void shiftnd_set_tuning(int knob, int value) {
    if (knob < 0 || knob >= kKnobs) return;
    g_knob[knob] = value;
    apply_knob(knob, value);
}

int shiftnd_debug_map(int64_t p, int64_t shift, int64_t len, int padding_mode) {
    return sweep_debug_map(p, shift, len, padding_mode);
}

// check_borders, ops/shifts.cpp:93-135 (host arithmetic only)
int shiftnd_check_borders(const int64_t *sizes, int nsizes, const int32_t *user, int ndim, int32_t borders[6],
                          int64_t *new_sizes) {
    if (!sizes || !borders || !new_sizes || ndim < 1 || nsizes < ndim + 1) return SHIFTND_ERR_INVALID_ARGUMENT;
    const int shift = ((ndim + 1) == nsizes) ? 1 : 2;
    const int hdim = 3;
    const int dims = ndim < hdim ? ndim : hdim;
    if (shift + dims > nsizes) return SHIFTND_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < hdim; ++i) {
        borders[2 * i] = 0;
        borders[2 * i + 1] = (i + 1 > ndim) ? 1 : static_cast<int32_t>(sizes[i + shift]);
    }
    if (user) {
        for (int i = 0; i < dims; ++i) {
            const int32_t size = static_cast<int32_t>(sizes[i + shift]);
            int32_t l = user[2 * i];
            int32_t r = borders[2 * i + 1] - user[2 * i + 1];
            if (r - l < 1) r = l + 1;          // degenerate window -> one element wide
            if (l == size) { l = size - 1; r = l + 1; }
            if (r == 0) { l = 0; r = 1; }
            if (l < 0) l = 0;
            if (r > size) r = size;
            borders[2 * i] = l;
            borders[2 * i + 1] = r;
        }
    }
    for (int i = 0; i < shift; ++i) new_sizes[i] = sizes[i];
    for (int i = 0; i < dims; ++i) new_sizes[i + shift] = static_cast<int64_t>(borders[2 * i + 1] - borders[2 * i]);
    return SHIFTND_OK;
}

int shiftnd_forward_serves_channels_last(const shiftnd_problem *p, const void *x, const int64_t x_strides[5], const void *out,
                                         const int64_t out_strides[5]) {
    if (!p || !x_strides || !out_strides || g_policy != 0) return 0;
    Geometry g;
    if (build_geometry(p, x_strides, out_strides, nullptr, g) != SHIFTND_OK || empty_problem(g)) return 0;
    // 16-bit interpolation: one layout change + the contiguous kernel is faster than the tiled kernel (N16 C256 224x224
    // bf16 through the op: 0.30 vs 0.54 ms; tools/cl_op_bench.py), so the answer is no although shiftnd_forward would
    // take a channels-last tensor
    if (g.active && (p->dtype == SHIFTND_F16 || p->dtype == SHIFTND_BF16)) return 0;
    // NDHWC (round 4): the tiled kernel stages element by element.  N8 C128 16x112x112 (tools/cl3d_bench.py): fp32 0.42 ms to an
    // NDHWC output, 0.49 to NCDHW, against 0.57 ms for the layout change + the contiguous kernel (+ a transpose back when the format
    // is kept); bf16 0.29 / 0.36 against 0.29 -- so 2-byte elements to an NCDHW output keep the layout change
    if (g.nd == 3 && !(g.os[1] == 1 && g.C > 1) && dtype_size(p->dtype) != 4) return 0;
    // ... and the interpolating NDHWC kernel (plane blend at staging time, fp32 ring) only wins for fp32 to an NDHWC output: 0.56 ms
    // against 0.59 + a transpose back (to NCDHW 0.61 against 0.59; bf16 0.49 - 0.52 against 0.29)
    if (g.nd == 3 && g.active && !(g.os[1] == 1 && g.C > 1)) return 0;
    return cl_tiled_forward_eligible(g, p->dtype, x, out) ? 1 : 0;
}

int shiftnd_backward_serves_channels_last(const shiftnd_problem *p, const void *grad_out, const int64_t grad_out_strides[5],
                                          const void *x, const int64_t x_strides[5], const void *grad_x,
                                          const int64_t grad_x_strides[5]) {
    if (!p || !grad_out_strides || !x_strides || !grad_x_strides || g_policy != 0 || !is_float_dtype(p->dtype)) return 0;
    Geometry g;
    if (build_geometry(p, x_strides, grad_out_strides, grad_x_strides, g) != SHIFTND_OK) return 0;
    if (g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0) return 0;
    if (g.nd == 3) {
        // NDHWC (round 5): shiftnd_cl_tiled3.hip serves such tensors as they lie (what a C-ABI caller gets: 1.2 - 1.5 ms on N8 C128
        // 16x112x112 fp32 where the channel-fastest kernels took 21 - 36 ms), but the op's route -- shiftnd_transpose of the saved
        // input, the walk kernels, shiftnd_transpose of grad_x back -- measures 1.04 ms (bf16 0.53 against 1.0 - 1.4;
        // tools/cl3d_bench.py): the answer stays no unless knob 23 asks for the direct kernel
        if (g_cl3 != 2) return 0;
        return cl_tiled3_backward_eligible(g, p->dtype, grad_out, x, grad_x) ? 1 : 0;
    }
    if (g.active && (p->dtype == SHIFTND_F16 || p->dtype == SHIFTND_BF16)) return 0;  // (0.56 vs 0.66 ms: see above)
    return cl_tiled_backward_eligible(g, p->dtype, grad_out, x, grad_x) ? 1 : 0;
}

int shiftnd_forward(const shiftnd_problem *p, const void *x, const int64_t x_strides[5], const void *weights, void *out,
                    const int64_t out_strides[5], void *stream) {
    if (!p || !x_strides || !out_strides) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!is_float_dtype(p->dtype)) return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    return forward_common(p, x, x_strides, weights, p->dtype, 0, 0ull, out, out_strides, stream);
}

int shiftnd_forward_quantized(const shiftnd_problem *p, const void *x, const int64_t x_strides[5], const void *wq,
                              int32_t wq_dtype, int64_t w_zero_point, int64_t x_zero_point, void *out,
                              const int64_t out_strides[5], void *stream) {
    if (!p || !x_strides || !out_strides) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!is_quant_dtype(p->dtype) || !is_quant_dtype(wq_dtype)) return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    shiftnd_problem q = *p;
    q.active = 0;
    // fill value = input zero point, as the element type's bit pattern
    uint64_t fill = 0;
    if (p->dtype == SHIFTND_I8) fill = static_cast<uint8_t>(static_cast<int8_t>(x_zero_point));
    else if (p->dtype == SHIFTND_U8) fill = static_cast<uint8_t>(x_zero_point);
    else fill = static_cast<uint32_t>(static_cast<int32_t>(x_zero_point));
    return forward_common(&q, x, x_strides, wq, wq_dtype, w_zero_point, fill, out, out_strides, stream);
}

// bytes of partial-sum records the backward families can need for this geometry under the calling thread's knobs
static size_t backward_workspace_now(const Geometry &g, int dtype) {
    size_t m = strided_backward_workspace(g);
    m = std::max(m, plane_backward_workspace(g, dtype));
    m = std::max(m, sweep_backward_workspace(g, dtype));
    m = std::max(m, cl_backward_workspace(g));
    m = std::max(m, cl_tiled_backward_workspace(g));
    m = std::max(m, small_backward_workspace(g, dtype));
    m = std::max(m, cl_tiled3_backward_workspace(g));
    m = std::max(m, flat_backward_workspace(g));
    m = std::max(m, plane_ragged_backward_workspace(g, dtype));
    return m;
}

size_t shiftnd_backward_workspace_bytes(const shiftnd_problem *p) {
    if (!p) return 0;
    // the layout of the partial-sum buffer depends only on the geometry: evaluate every family's plan
    const int64_t unit[5] = {0, 0, 0, 0, 0};
    Geometry g;
    if (build_geometry(p, unit, unit, unit, g) != SHIFTND_OK) return 0;
    if (g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0) return sizeof(double);
    size_t m = backward_workspace_now(g, p->dtype);
    if (knobs_touched()) {   // ... and under the default knobs: what a run on any thread can fall back to
        DefaultKnobs scope;
        m = std::max(m, backward_workspace_now(g, p->dtype));
    }
    return m;
}

static int backward_planned(const shiftnd_problem *p, const void *grad_out, const int64_t grad_out_strides[5], const void *x,
                            const int64_t x_strides[5], const void *weights, void *grad_x, const int64_t grad_x_strides[5],
                            void *grad_w, void *workspace, size_t workspace_bytes, void *stream);

int shiftnd_backward(const shiftnd_problem *p, const void *grad_out, const int64_t grad_out_strides[5], const void *x,
                     const int64_t x_strides[5], const void *weights, void *grad_x, const int64_t grad_x_strides[5],
                     void *grad_w, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = backward_planned(p, grad_out, grad_out_strides, x, x_strides, weights, grad_x, grad_x_strides, grad_w, workspace,
                              workspace_bytes, stream);
    if (rc == SHIFTND_ERR_WORKSPACE_TOO_SMALL && knobs_touched()) {
        // (nothing was launched.)  This thread's knobs plan more partial records than the caller sized for -- e.g. sized on
        // a thread with the default knobs: run the default plan, which shiftnd_backward_workspace_bytes always covers
        DefaultKnobs scope;
        rc = backward_planned(p, grad_out, grad_out_strides, x, x_strides, weights, grad_x, grad_x_strides, grad_w, workspace,
                              workspace_bytes, stream);
    }
    return rc;
}

static int backward_planned(const shiftnd_problem *p, const void *grad_out, const int64_t grad_out_strides[5], const void *x,
                            const int64_t x_strides[5], const void *weights, void *grad_x, const int64_t grad_x_strides[5],
                            void *grad_w, void *workspace, size_t workspace_bytes, void *stream) {
    if (!p || !grad_out_strides || !x_strides || !grad_x_strides) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!is_float_dtype(p->dtype)) return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    Geometry g;
    const int rc = build_geometry(p, x_strides, grad_out_strides, grad_x_strides, g);
    if (rc != SHIFTND_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0) {
        // nothing to differentiate; grad_w (if any channels) is all zeros (zeros_like, shifts_cpu.cpp:247)
        g_last_path = SHIFTND_PATH_EMPTY;
        if (g.C > 0 && grad_w)
            if (hipMemsetAsync(grad_w, 0, static_cast<size_t>(g.C) * g.nd * dtype_size(p->dtype), st) != hipSuccess)
                return SHIFTND_ERR_LAUNCH_FAILED;
        return SHIFTND_OK;
    }
    if (!grad_out || !x || !weights || !grad_x || !grad_w || !workspace) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (g_policy == 0 && g_flat == 2 && flat_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {   // (knob 27 = 2: tests)
        if (flat_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(flat_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    const bool can_sweep = sweep_backward_eligible(g, p->dtype, grad_out, x, grad_x);
    const bool can_plane = plane_backward_eligible(g, p->dtype, grad_out, x, grad_x);
    if ((g_policy == 2 && !can_plane) || (g_policy == 3 && !can_sweep)) return SHIFTND_ERR_INVALID_ARGUMENT;
    // automatic choice: the per-plane kernels (faster for the backward pass); the sweep kernels take over
    // when the index maps do not fit in LDS
    if (can_sweep && (g_policy == 3 || (g_policy == 0 && !can_plane))) {
        if (sweep_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_SWEEP;
        return finish(sweep_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    // ragged rows of 4- / 8-byte elements on planes with rows of at least 8 chunks: the row-relative crop_backward (shiftnd_span.hip, XRAG)
    if (g_policy == 0 && g_flat != 2 && ragged_rows(g, p->dtype) && span_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {
        if (span_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(span_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    // ragged input rows -- and windows crop_backward does not take: the flat chunk stream.  (Ragged rows of 4- / 8-byte elements on
    // planes large enough for the row-relative crop_backward: that one, through plane_backward below.)
    if (g_policy == 0 && !span_backward_eligible(g, p->dtype, grad_out, x, grad_x) && (ragged_rows(g, p->dtype) || cropped(g)) &&
        flat_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {
        if (flat_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(flat_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if (can_plane && g_policy != 1 && g_policy != 4) {
        if (plane_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(plane_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if (g_policy == 0 && small_backward_eligible(g, p->dtype)) {  // rows not whole 16-byte pieces, small planes
        if (small_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(small_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    // 3-D volumes with ragged rows beyond the small-plane kernels (16 x 28 x 28 bf16, 8 x 56 x 62 fp32 ...): the direct-load plane
    // kernels with 4- / 8-byte chunks instead of the one-thread-per-element fallback
    if (g_policy == 0 && plane_ragged_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {
        if (plane_ragged_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(plane_ragged_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if ((g_policy == 0 || g_policy == 4) && cl_tiled_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {  // all channels-last: LDS-tiled
        if (cl_tiled_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_CL;
        return finish(cl_tiled_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if ((g_policy == 0 || g_policy == 4) && g_cl3 != 1 && cl_tiled3_backward_eligible(g, p->dtype, grad_out, x, grad_x)) {  // NDHWC: LDS-tiled, one plane per workgroup
        if (cl_tiled3_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_CL;
        return finish(cl_tiled3_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if (g_policy == 4 && !cl_backward_eligible(g, p->dtype)) return SHIFTND_ERR_INVALID_ARGUMENT;
    if ((g_policy == 0 && cl_backward_preferred(g, p->dtype)) || g_policy == 4) {
        if (cl_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_CL;
        return finish(cl_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
    }
    if (strided_backward_workspace(g) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
    g_last_path = SHIFTND_PATH_STRIDED;
    return finish(strided_backward(g, p->dtype, grad_out, x, weights, grad_x, grad_w, workspace, st));
}

// ---- layout change ---------------------------------------------------------------------------------------------
int shiftnd_transpose(const void *src, void *dst, int64_t batch, int64_t rows, int64_t cols, int32_t element_bytes,
                      void *stream) {
    if (batch < 0 || rows < 0 || cols < 0) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (batch == 0 || rows == 0 || cols == 0) return SHIFTND_OK;
    if (!src || !dst) return SHIFTND_ERR_INVALID_ARGUMENT;
    return finish(transpose_planes(src, dst, batch, rows, cols, element_bytes, static_cast<hipStream_t>(stream)));
}

// ---- fused shift + average pool ----------------------------------------------------------------------------
static int pooled_geometry(const shiftnd_problem *p, const int32_t *pool, Geometry &g, bool quantized_too = false) {
    if (!p || !pool) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!is_float_dtype(p->dtype) && !(quantized_too && is_quant_dtype(p->dtype))) return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    const int64_t unit[5] = {0, 0, 0, 0, 0};
    const int rc = build_geometry(p, unit, unit, unit, g);
    if (rc != SHIFTND_OK) return rc;
    const int lead = 3 - p->ndim;
    for (int d = 0; d < 3; ++d) g.K[d] = 1;
    for (int r = 0; r < p->ndim; ++r) {
        if (pool[r] < 1) return SHIFTND_ERR_INVALID_ARGUMENT;
        g.K[r + lead] = pool[r];
    }
    for (int d = 0; d < 3; ++d) g.P[d] = (g.O[d] + g.K[d] - 1) / g.K[d];
    // contiguous tensors: element strides in normalised order N, C, d0, d1, inner
    auto fill = [&](int64_t *st, const int64_t *sz) {
        st[4] = 1;
        st[3] = sz[2];
        st[2] = sz[2] * sz[1];
        st[1] = sz[2] * sz[1] * sz[0];
        st[0] = st[1] * g.C;
    };
    fill(g.xs, g.S);
    fill(g.gs, g.S);
    fill(g.os, g.P);
    return SHIFTND_OK;
}

size_t shiftnd_backward_pooled_workspace_bytes(const shiftnd_problem *p, const int32_t *pool) {
    // the pooled backward plans its launch with the pool in the geometry (more, shorter workgroups than the plain
    // backward of the same tensor can need more partial-sum groups)
    Geometry g;
    if (pooled_geometry(p, pool, g) != SHIFTND_OK) return 0;
    if (g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0) return sizeof(double);
    size_t m = plane_backward_workspace(g, p->dtype);
    if (knobs_touched()) {   // (as shiftnd_backward_workspace_bytes)
        DefaultKnobs scope;
        m = std::max(m, plane_backward_workspace(g, p->dtype));
    }
    return m;
}

int shiftnd_pooled_sizes(const shiftnd_problem *p, const int32_t *pool, int64_t pooled_spatial[3]) {
    Geometry g;
    const int rc = pooled_geometry(p, pool, g, true);  // (the sizes do not depend on the element type)
    if (rc != SHIFTND_OK) return rc;
    if (!pooled_spatial) return SHIFTND_ERR_INVALID_ARGUMENT;
    const int lead = 3 - p->ndim;
    for (int r = 0; r < 3; ++r) pooled_spatial[r] = r < p->ndim ? g.P[r + lead] : 1;
    return SHIFTND_OK;
}

int shiftnd_forward_pooled(const shiftnd_problem *p, const int32_t *pool, const void *x, const void *weights, void *out,
                           void *stream) {
    Geometry g;
    const int rc = pooled_geometry(p, pool, g);
    if (rc != SHIFTND_OK) return rc;
    if (empty_problem(g)) {
        g_last_path = SHIFTND_PATH_EMPTY;
        return SHIFTND_OK;
    }
    if (!x || !weights || !out) return SHIFTND_ERR_INVALID_ARGUMENT;
    // 3-D interpolating: the walk through the planes with the pool as its epilogue
    if (g_policy == 0 && walk_forward_pooled_eligible(g, p->dtype, x, out)) {
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(walk_forward(g, p->dtype, x, weights, p->dtype, out, static_cast<hipStream_t>(stream)));
    }
    // 2-D sparse shift, 2 x 2 windows, fp32 / fp64: the linear sweep of one-step workgroups with the pool as its epilogue
    if (g_policy == 0 && step_forward_pooled_eligible(g, p->dtype, x, out)) {
        g_last_path = SHIFTND_PATH_SWEEP;
        return finish(step_forward_pooled(g, p->dtype, x, weights, p->dtype, out, static_cast<hipStream_t>(stream)));
    }
    // 3-D volumes, 2 x 2 x 2 windows, cropped or not (what the walk above does not take): pooled rows through LDS (round 6)
    if (g_policy == 0 && span_forward_pooled3_eligible(g, p->dtype, x, out)) {
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(span_forward_pooled3(g, p->dtype, x, weights, p->dtype, out, static_cast<hipStream_t>(stream)));
    }
    // 1-D rows of at least 128 chunks, windows of 2: row_forward with the pool as its epilogue (round 6)
    if (g_policy == 0 && span_forward_pooled_eligible(g, p->dtype, x, out)) {
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(span_forward_pooled(g, p->dtype, x, weights, p->dtype, out, static_cast<hipStream_t>(stream)));
    }
    if (!plane_pool_forward_eligible(g, p->dtype)) return SHIFTND_ERR_NOT_FUSED;
    g_last_path = SHIFTND_PATH_PLANE;
    return finish(plane_pool_forward(g, p->dtype, x, weights, out, static_cast<hipStream_t>(stream)));
}

int shiftnd_forward_quantized_pooled(const shiftnd_problem *p, const int32_t *pool, const void *x, const void *wq, int32_t wq_dtype,
                                     int64_t w_zero_point, int64_t x_zero_point, int32_t requant, void *out, void *stream) {
    if (!p || !pool) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (requant != SHIFTND_REQUANT_ZP_INSIDE && requant != SHIFTND_REQUANT_ZP_OUTSIDE) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!is_quant_dtype(p->dtype) || !is_quant_dtype(wq_dtype)) return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    shiftnd_problem q = *p;
    q.active = 0;
    Geometry g;
    const int rc = pooled_geometry(&q, pool, g, true);
    if (rc != SHIFTND_OK) return rc;
    if (empty_problem(g)) {
        g_last_path = SHIFTND_PATH_EMPTY;
        return SHIFTND_OK;
    }
    if (!x || !wq || !out) return SHIFTND_ERR_INVALID_ARGUMENT;
    if (!qpool_forward_eligible(g, p->dtype)) return SHIFTND_ERR_NOT_FUSED;
    g_last_path = SHIFTND_PATH_PLANE;
    return finish(qpool_forward(g, p->dtype, x, wq, wq_dtype, w_zero_point, x_zero_point, requant, out, static_cast<hipStream_t>(stream)));
}

static int backward_pooled_planned(const shiftnd_problem *p, const int32_t *pool, const void *grad_pooled, const void *x,
                                   const void *weights, void *grad_x, void *grad_w, void *workspace, size_t workspace_bytes,
                                   void *stream);

int shiftnd_backward_pooled(const shiftnd_problem *p, const int32_t *pool, const void *grad_pooled, const void *x,
                            const void *weights, void *grad_x, void *grad_w, void *workspace, size_t workspace_bytes,
                            void *stream) {
    int rc = backward_pooled_planned(p, pool, grad_pooled, x, weights, grad_x, grad_w, workspace, workspace_bytes, stream);
    if (rc == SHIFTND_ERR_WORKSPACE_TOO_SMALL && knobs_touched()) {   // (as shiftnd_backward)
        DefaultKnobs scope;
        rc = backward_pooled_planned(p, pool, grad_pooled, x, weights, grad_x, grad_w, workspace, workspace_bytes, stream);
    }
    return rc;
}

static int backward_pooled_planned(const shiftnd_problem *p, const int32_t *pool, const void *grad_pooled, const void *x,
                                   const void *weights, void *grad_x, void *grad_w, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    Geometry g;
    const int rc = pooled_geometry(p, pool, g);
    if (rc != SHIFTND_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g.N == 0 || g.C == 0 || g.S[0] * g.S[1] * g.S[2] == 0) {
        g_last_path = SHIFTND_PATH_EMPTY;
        if (g.C > 0 && grad_w)
            if (hipMemsetAsync(grad_w, 0, static_cast<size_t>(g.C) * g.nd * dtype_size(p->dtype), st) != hipSuccess)
                return SHIFTND_ERR_LAUNCH_FAILED;
        return SHIFTND_OK;
    }
    if (!grad_pooled || !x || !weights || !grad_x || !grad_w || !workspace) return SHIFTND_ERR_INVALID_ARGUMENT;
    // 3-D interpolating: the walk through the planes with the pooled gradient expanded on the way into LDS
    if (g_policy == 0 && walk_backward_pooled_eligible(g, p->dtype, grad_pooled, x, grad_x)) {
        if (plane_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(step_backward(g, p->dtype, grad_pooled, x, weights, grad_x, grad_w, workspace, st));
    }
    // cropped 3-D volumes, 2 x 2 x 2 windows: crop_backward3 with the pooled gradient expanded on its way into LDS (round 6)
    if (g_policy == 0 && g.nd == 3 && span_backward_pooled_eligible(g, p->dtype, grad_pooled, x, grad_x)) {
        if (span_backward_pooled_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
        g_last_path = SHIFTND_PATH_PLANE;
        return finish(span_backward(g, p->dtype, grad_pooled, x, weights, grad_x, grad_w, workspace, st));
    }
    if (!plane_pool_backward_eligible(g, p->dtype, grad_x)) return SHIFTND_ERR_NOT_FUSED;
    // what the walk does not take of the 3-D interpolating backward: through the band-walk kernels 16 gradient corner rows per
    // step would be expanded from pooled rows -- measured slower than avg_pool backward + shiftnd_backward (N8 C128 16x112x112:
    // 1.42 vs 1.33 ms fp32, 1.32 vs 0.96 ms bf16), so it is not fused unless the plane-kernel policy is forced (tests)
    if (g.nd == 3 && g.active && g_policy != 2) return SHIFTND_ERR_NOT_FUSED;
    // (round 6) a cropped 3-D volume that crop_backward3 serves: the band-walk kernels with the pool fused are SLOWER than the average
    // pool's backward followed by crop_backward3 (N8 C128 16x56x56 cut 1/1/1 pool 2, sparse: bf16 0.32 vs 0.10 + the pool, fp32 0.33 vs
    // 0.17 + the pool; tools/module_matrix.py) -- not fused, the op composes the two
    if (g.nd == 3 && g_policy == 0) {
        Geometry u = g;   // the unpooled problem: gradient of the window, dense
        for (int d = 0; d < 3; ++d) u.K[d] = 0, u.P[d] = 0;
        u.os[4] = 1;
        u.os[3] = u.O[2];
        u.os[2] = u.O[2] * u.O[1];
        u.os[1] = u.O[2] * u.O[1] * u.O[0];
        u.os[0] = u.os[1] * u.C;
        if (span_backward_eligible(u, p->dtype, x, x, grad_x)) return SHIFTND_ERR_NOT_FUSED;   // (the gradient will be a fresh, aligned tensor)
    }
    if (plane_backward_workspace(g, p->dtype) > workspace_bytes) return SHIFTND_ERR_WORKSPACE_TOO_SMALL;
    g_last_path = SHIFTND_PATH_PLANE;
    return finish(plane_pool_backward(g, p->dtype, grad_pooled, x, weights, grad_x, grad_w, workspace, st));
}

}  // extern "C"
