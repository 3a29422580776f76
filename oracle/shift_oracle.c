/*
 * shift_oracle.c -- CPU ORACLE for the shiftnd hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C, single-threaded restatement of the reference's shiftnd
 * algorithm (DeadAt0m/ActiveSparseShifts-PyTorch, torchshifts/csrc).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the
 * HIP kernels; NOTHING in the product path (activesparseshifts-pytorch_amd/) may
 * import, link or call it.
 *
 * Parity status: PINNED.  The reference ships no golden vectors (tests/shifts_test.py
 * only asserts `grad is not None`), so this oracle is pinned against outputs of the
 * reference itself: oracle/build_ref.sh compiles the reference's own C++ CPU path from
 * /root/reference into oracle/_ref/, tests/golden/make_golden.py records its outputs,
 * and tests/test_oracle_golden.py requires this file to reproduce them bit for bit
 * (fp32/fp64 forward, input-grad, weight-grad in reference summation order, quantized).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/torchshifts/csrc/ops/).
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared shift_oracle.c -o liboracle.so -lm
 * (-ffp-contract=off: the reference is built by g++ -O3 for baseline x86-64, i.e. with
 *  separate multiply and add; an FMA would change the last bit of interp1D.)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef int64_t idx_t;

enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_PERIODIC = 2, PAD_REFLECT = 3, PAD_SYMMETRIC = 4 };

/* kernels/shifts_kernels.h:7-8  mod(a,b) = (b + (a % b)) % b */
static inline idx_t mod_(idx_t a, idx_t b) { return (b + (a % b)) % b; }

static inline idx_t iabs_(idx_t a) { return a < 0 ? -a : a; }

/* kernels/shifts_kernels.h:10-29  infer_index<idx_t, padding_mode> */
idx_t oracle_infer_index(idx_t index, idx_t len, int pad)
{
    int odd;
    switch (pad) {
    case PAD_ZEROS:
        return (index > len - 1) ? -1 : index; /* negatives pass through, rejected by >=0 later */
    case PAD_BORDER: {
        idx_t t = index > 0 ? index : 0;
        return (len - 1 < t) ? len - 1 : t;
    }
    case PAD_PERIODIC:
        return mod_(index, len);
    case PAD_REFLECT:
        odd = (int)((((idx_t)(index < 0)) + (iabs_(index) - (idx_t)(index < 0)) / (len - 1)) & 1);
        return odd ? (len - 1 - mod_(index, len - 1)) : mod_(index, len - 1);
    case PAD_SYMMETRIC:
        odd = (int)((((idx_t)(index < 0)) + (iabs_(index) - (idx_t)(index < 0)) / len) & 1);
        return odd ? (len - 1 - mod_(index, len)) : mod_(index, len);
    default:
        return (index > len - 1) ? -1 : index;
    }
}

/* Problem geometry shared by all entry points (element strides, reference naming H,W,D). */
typedef struct {
    int32_t nd;           /* spatial dims 1..3 */
    int32_t pad;          /* 0..4 */
    int32_t active;       /* 0/1 */
    int32_t nhwc_order;   /* 0: loop (n,c,i,j,k) cpu/shifts_cpu.cpp:78-98; 1: (n,i,j,k,c) :57-75 */
    idx_t N, C, H, W, D;  /* input sizes; unused dims = 1 */
    idx_t x_s[5];         /* input strides N,C,H,W,D (0 for unused dims) */
    idx_t o_s[5];         /* forward: output strides; backward: grad_out (the incoming grad) strides */
    idx_t gx_s[5];        /* backward only: grad_x strides */
    idx_t b[6];           /* borders l_i,r_i,l_j,r_j,l_k,r_k (cpu/shifts_cpu.cpp:46-52) */
} oracle_geom;

/* kernels/shifts_kernels.h:32-54 get_shifted_value: the three per-dim index resolutions.
 * Returns the element offset or -1 when the gather is masked out. */
static inline idx_t gather_offset(const oracle_geom *g, int pad,
                                  idx_t is, idx_t sizeH, idx_t sH,
                                  idx_t js, idx_t sizeW, idx_t sW,
                                  idx_t ks, idx_t sizeD, idx_t sD,
                                  idx_t c, idx_t sC, int out_passcond)
{
    const int nd = g->nd;
    const idx_t ti = (sizeH == 1) ? 0 : oracle_infer_index(is, sizeH, pad);
    const idx_t pi = (idx_t)(ti >= 0);
    const idx_t oH = ti * sH * pi;
    const idx_t tj = (nd > 1) ? ((sizeW == 1) ? 0 : oracle_infer_index(js, sizeW, pad)) : 0;
    const idx_t pj = (nd > 1) ? ((idx_t)(tj >= 0) * pi) : pi;
    const idx_t oW = (nd > 1) ? tj * sW * pj : 0;
    const idx_t tk = (nd > 2) ? ((sizeD == 1) ? 0 : oracle_infer_index(ks, sizeD, pad)) : 0;
    const idx_t pk = (nd > 2) ? ((idx_t)(tk >= 0) * pj) : pj;
    const idx_t oD = (nd > 2) ? tk * sD * pk : 0;
    if (pk && out_passcond) return oH + oW + oD + c * sC;
    return -1;
}

/* ---------------------------------------------------------------------------------------------
 * Floating-point paths, instantiated for float and double.
 * ------------------------------------------------------------------------------------------- */
#define DEFINE_FLOAT_ORACLE(T, SUF, ROUND, FLOOR, CEIL)                                          \
                                                                                                 \
/* kernels/interpolation.h:3-61 */                                                               \
static inline T interp1D_##SUF(T v1, T v2, T x) { return v1 * (1 - x) + v2 * x; }                \
static inline T interp1D_dx_##SUF(T v1, T v2) { return v2 - v1; }                                \
static inline T interp2D_##SUF(T v1, T v2, T v3, T v4, T x, T y)                                 \
{ return interp1D_##SUF(interp1D_##SUF(v1, v2, x), interp1D_##SUF(v3, v4, x), y); }              \
static inline T interp2D_dx_##SUF(T v1, T v2, T v3, T v4, T y)                                   \
{ return interp1D_##SUF(interp1D_dx_##SUF(v1, v3), interp1D_dx_##SUF(v2, v4), y); }              \
static inline T interp2D_dy_##SUF(T v1, T v2, T v3, T v4, T x)                                   \
{ return interp1D_dx_##SUF(interp1D_##SUF(v1, v2, x), interp1D_##SUF(v3, v4, x)); }              \
static inline T interp3D_##SUF(const T *v, T x, T y, T z)                                        \
{ return interp1D_##SUF(interp2D_##SUF(v[0], v[1], v[2], v[3], x, y),                            \
                        interp2D_##SUF(v[4], v[5], v[6], v[7], x, y), z); }                      \
static inline T interp3D_dx_##SUF(const T *v, T y, T z)                                          \
{ return interp1D_##SUF(interp2D_dx_##SUF(v[0], v[1], v[2], v[3], y),                            \
                        interp2D_dx_##SUF(v[4], v[5], v[6], v[7], y), z); }                      \
static inline T interp3D_dy_##SUF(const T *v, T x, T z)                                          \
{ return interp1D_##SUF(interp2D_dy_##SUF(v[0], v[1], v[2], v[3], x),                            \
                        interp2D_dy_##SUF(v[4], v[5], v[6], v[7], x), z); }                      \
static inline T interp3D_dz_##SUF(const T *v, T x, T y)                                          \
{ return interp1D_dx_##SUF(interp2D_##SUF(v[0], v[1], v[2], v[3], x, y),                         \
                           interp2D_##SUF(v[4], v[5], v[6], v[7], x, y)); }                      \
                                                                                                 \
/* kernels/shifts_kernels.h:58-103 get_shifted_values: corner order                              \
 * v0=(i,j,k) v1=(i+1,j,k) v2=(i,j+1,k) v3=(i+1,j+1,k) v4..7 = same with k+1 */                  \
static inline void corners_##SUF(const oracle_geom *g, const T *arr,                             \
                                 idx_t is, idx_t sizeH, idx_t sH,                                \
                                 idx_t js, idx_t sizeW, idx_t sW,                                \
                                 idx_t ks, idx_t sizeD, idx_t sD,                                \
                                 idx_t c, idx_t sC, int passcond, T *v)                          \
{                                                                                                \
    const int ncorner = 1 << g->nd;                                                              \
    for (int q = 0; q < ncorner; ++q) {                                                          \
        idx_t off = gather_offset(g, g->pad, is + (q & 1), sizeH, sH,                            \
                                  js + ((q >> 1) & 1), sizeW, sW,                                \
                                  ks + ((q >> 2) & 1), sizeD, sD, c, sC, passcond);              \
        v[q] = (off >= 0) ? arr[off] : (T)0;                                                     \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* kernels/shifts_kernels.h:110-130 compute_interpolated (reverse=false is the only use) */      \
static inline T interpolated_##SUF(int nd, const T *v, T dH, T dW, T dD, int pass)               \
{                                                                                                \
    if (!pass) return (T)0;                                                                      \
    if (nd == 3) return interp3D_##SUF(v, dH, dW, dD);                                           \
    if (nd == 2) return interp2D_##SUF(v[0], v[1], v[2], v[3], dH, dW);                          \
    return interp1D_##SUF(v[0], v[1], dH);                                                       \
}                                                                                                \
                                                                                                 \
/* kernels/shifts_kernels.h:132-154 compute_weight_gradients (mis-wired 2-D/3-D dx kept) */      \
static inline void weight_grads_##SUF(int nd, const T *v, T dH, T dW, T dD, int pass, T *o)      \
{                                                                                                \
    if (nd == 3) {                                                                               \
        o[0] = pass ? interp3D_dx_##SUF(v, dW, dD) : (T)0;                                       \
        o[1] = pass ? interp3D_dy_##SUF(v, dH, dD) : (T)0;                                       \
        o[2] = pass ? interp3D_dz_##SUF(v, dH, dW) : (T)0;                                       \
    } else if (nd == 2) {                                                                        \
        o[0] = pass ? interp2D_dx_##SUF(v[0], v[1], v[2], v[3], dW) : (T)0;                      \
        o[1] = pass ? interp2D_dy_##SUF(v[0], v[1], v[2], v[3], dH) : (T)0;                      \
    } else {                                                                                     \
        o[0] = pass ? interp1D_dx_##SUF(v[0], v[1]) : (T)0;                                      \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* cpu/shifts_cpu.cpp:223-224 forward weight prep.                                               \
 * torch::round is round-half-to-even == rint under the default rounding mode. */                \
void oracle_weights_forward_##SUF(const T *w, idx_t n, int active, idx_t *iw, T *dw)             \
{                                                                                                \
    for (idx_t t = 0; t < n; ++t) {                                                              \
        T r = active ? FLOOR(w[t]) : ROUND(w[t]);                                                \
        iw[t] = (idx_t)r;                        /* .to(torch::kLong) truncates */               \
        dw[t] = active ? (T)(w[t] - (T)iw[t]) : (T)0;                                            \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* cpu/shifts_cpu.cpp:242-244 backward weight prep */                                            \
void oracle_weights_backward_##SUF(const T *w, idx_t n, int active, idx_t *iw, T *dw)            \
{                                                                                                \
    for (idx_t t = 0; t < n; ++t) {                                                              \
        T d = active ? (T)(w[t] - FLOOR(w[t]))                                                   \
                     : ((w[t] > 0) ? (T)(w[t] - FLOOR(w[t])) : (T)(CEIL(w[t]) - w[t]));          \
        dw[t] = d;                                                                               \
        T r = active ? (T)(w[t] - d) : ROUND(w[t]);                                              \
        iw[t] = (idx_t)r;                                                                        \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* kernels/shifts_kernels.h:156-220 shift_forward_kernel_nchwd (and :330-400, same math) */      \
static inline void fwd_elem_##SUF(const oracle_geom *g, const T *x, T *out,                      \
                                  const idx_t *iw, const T *dw,                                  \
                                  idx_t n, idx_t c, idx_t i, idx_t j, idx_t k)                   \
{                                                                                                \
    const int nd = g->nd;                                                                        \
    const idx_t li = g->b[0], ri = g->b[1];                                                      \
    const idx_t lj = nd < 2 ? 0 : g->b[2], rj = nd < 2 ? 1 : g->b[3];                            \
    const idx_t lk = nd < 3 ? 0 : g->b[4], rk = nd < 3 ? 1 : g->b[5];                            \
    const int pass = (i >= li) && (i < ri) && (j >= lj) && (j < rj) && (k >= lk) && (k < rk);    \
    if (!pass) return;                                                                           \
    const idx_t oi = i - li, oj = nd > 1 ? j - lj : j, ok = nd > 2 ? k - lk : k;                 \
    const idx_t si = i - iw[c * nd];                                                             \
    const idx_t sj = nd > 1 ? j - iw[c * nd + 1] : j;                                            \
    const idx_t sk = nd > 2 ? k - iw[c * nd + 2] : k;                                            \
    const T *xn = x + n * g->x_s[0];                                                             \
    T *o = out + n * g->o_s[0] + c * g->o_s[1] + oi * g->o_s[2] + oj * g->o_s[3] + ok * g->o_s[4];\
    if (g->active) {                                                                             \
        T v[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                       \
        const T dH = dw[c * nd];                                                                 \
        const T dW = nd > 1 ? dw[c * nd + 1] : (T)0;                                             \
        const T dD = nd > 2 ? dw[c * nd + 2] : (T)0;                                             \
        corners_##SUF(g, xn, si, g->H, g->x_s[2], sj, g->W, g->x_s[3], sk, g->D, g->x_s[4],      \
                      c, g->x_s[1], 1, v);                                                       \
        *o = interpolated_##SUF(nd, v, dH, dW, dD, 1);                                           \
    } else {                                                                                     \
        idx_t off = gather_offset(g, g->pad, si, g->H, g->x_s[2], sj, g->W, g->x_s[3],           \
                                  sk, g->D, g->x_s[4], c, g->x_s[1], 1);                         \
        *o = (off >= 0) ? xn[off] : (T)0;                                                        \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* cpu/shifts_cpu.cpp:18-100 shiftnd_forward_kernel loop nests */                                \
void oracle_forward_##SUF(const oracle_geom *g, const T *x, const idx_t *iw, const T *dw, T *out)\
{                                                                                                \
    if (g->nhwc_order) {                                                                         \
        for (idx_t n = 0; n < g->N; ++n)                                                         \
            for (idx_t i = 0; i < g->H; ++i)                                                     \
                for (idx_t j = 0; j < g->W; ++j)                                                 \
                    for (idx_t k = 0; k < g->D; ++k)                                             \
                        for (idx_t c = 0; c < g->C; ++c)                                         \
                            fwd_elem_##SUF(g, x, out, iw, dw, n, c, i, j, k);                    \
    } else {                                                                                     \
        for (idx_t n = 0; n < g->N; ++n)                                                         \
            for (idx_t c = 0; c < g->C; ++c)                                                     \
                for (idx_t i = 0; i < g->H; ++i)                                                 \
                    for (idx_t j = 0; j < g->W; ++j)                                             \
                        for (idx_t k = 0; k < g->D; ++k)                                         \
                            fwd_elem_##SUF(g, x, out, iw, dw, n, c, i, j, k);                    \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* kernels/shifts_kernels.h:222-327 shift_backward_kernel_nchwd (and :402-527).                  \
 * gw accumulates with a plain += in scalar_t (global_scope.h:22), in loop order. */             \
static inline void bwd_elem_##SUF(const oracle_geom *g, const T *go, const T *x, T *gx,          \
                                  const idx_t *iw, const T *dw, T *gw,                           \
                                  idx_t n, idx_t c, idx_t i, idx_t j, idx_t k)                   \
{                                                                                                \
    const int nd = g->nd;                                                                        \
    const idx_t li = g->b[0], ri = g->b[1];                                                      \
    const idx_t lj = nd < 2 ? 0 : g->b[2], rj = nd < 2 ? 1 : g->b[3];                            \
    const idx_t lk = nd < 3 ? 0 : g->b[4], rk = nd < 3 ? 1 : g->b[5];                            \
    const int pass = (i >= li) && (i < ri) && (j >= lj) && (j < rj) && (k >= lk) && (k < rk);    \
    const idx_t shi = iw[c * nd];                                                                \
    const idx_t shj = nd > 1 ? iw[c * nd + 1] : 0;                                               \
    const idx_t shk = nd > 2 ? iw[c * nd + 2] : 0;                                               \
    const T dH = dw[c * nd];                                                                     \
    const T dW = nd > 1 ? dw[c * nd + 1] : (T)0;                                                 \
    const T dD = nd > 2 ? dw[c * nd + 2] : (T)0;                                                 \
    const idx_t si = i - shi, sj = nd > 1 ? j - shj : j, sk = nd > 2 ? k - shk : k;              \
    const idx_t oi = i - li, oj = nd > 1 ? j - lj : j, ok = nd > 2 ? k - lk : k;                 \
    const T *gon = go + n * g->o_s[0];                                                           \
    const T *xn = x + n * g->x_s[0];                                                             \
    T v[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                           \
    T wg[3] = {0, 0, 0};                                                                         \
    const T gval = pass ? gon[c * g->o_s[1] + oi * g->o_s[2] + oj * g->o_s[3] + ok * g->o_s[4]]  \
                        : (T)0;                                                                  \
    corners_##SUF(g, xn, si, g->H, g->x_s[2], sj, g->W, g->x_s[3], sk, g->D, g->x_s[4],          \
                  c, g->x_s[1], pass, v);                                                        \
    weight_grads_##SUF(nd, v, dH, dW, dD, pass, wg);                                             \
    gw[c * nd] += gval * wg[0];                                                                  \
    if (nd > 1) gw[c * nd + 1] += gval * wg[1];                                                  \
    if (nd > 2) gw[c * nd + 2] += gval * wg[2];                                                  \
    const idx_t osH = ri - li, osW = rj - lj, osD = rk - lk;                                     \
    T *o = gx + n * g->gx_s[0] + c * g->gx_s[1] + i * g->gx_s[2] + j * g->gx_s[3] + k * g->gx_s[4];\
    if (g->active) {                                                                             \
        const idx_t osi = oi - shi, osj = nd > 1 ? oj - shj : oj, osk = nd > 2 ? ok - shk : ok;  \
        corners_##SUF(g, gon, osi, osH, g->o_s[2], osj, osW, g->o_s[3], osk, osD, g->o_s[4],     \
                      c, g->o_s[1], pass, v);                                                    \
        *o = interpolated_##SUF(nd, v, dH, dW, dD, pass);                                        \
    } else {                                                                                     \
        const idx_t rsi = oi + shi, rsj = nd > 1 ? oj + shj : oj, rsk = nd > 2 ? ok + shk : ok;  \
        idx_t off = gather_offset(g, g->pad, rsi, osH, g->o_s[2], rsj, osW, g->o_s[3],           \
                                  rsk, osD, g->o_s[4], c, g->o_s[1], pass);                      \
        *o = (off >= 0) ? gon[off] : (T)0;                                                       \
    }                                                                                            \
}                                                                                                \
                                                                                                 \
/* cpu/shifts_cpu.cpp:106-211 shiftnd_backward_kernel loop nests; gw must be zeroed by caller    \
 * (zeros_like at cpu/shifts_cpu.cpp:247). */                                                    \
void oracle_backward_##SUF(const oracle_geom *g, const T *go, const T *x,                        \
                           const idx_t *iw, const T *dw, T *gx, T *gw)                           \
{                                                                                                \
    if (g->nhwc_order) {                                                                         \
        for (idx_t n = 0; n < g->N; ++n)                                                         \
            for (idx_t i = 0; i < g->H; ++i)                                                     \
                for (idx_t j = 0; j < g->W; ++j)                                                 \
                    for (idx_t k = 0; k < g->D; ++k)                                             \
                        for (idx_t c = 0; c < g->C; ++c)                                         \
                            bwd_elem_##SUF(g, go, x, gx, iw, dw, gw, n, c, i, j, k);             \
    } else {                                                                                     \
        for (idx_t n = 0; n < g->N; ++n)                                                         \
            for (idx_t c = 0; c < g->C; ++c)                                                     \
                for (idx_t i = 0; i < g->H; ++i)                                                 \
                    for (idx_t j = 0; j < g->W; ++j)                                             \
                        for (idx_t k = 0; k < g->D; ++k)                                         \
                            bwd_elem_##SUF(g, go, x, gx, iw, dw, gw, n, c, i, j, k);             \
    }                                                                                            \
}

DEFINE_FLOAT_ORACLE(float, f32, rintf, floorf, ceilf)
DEFINE_FLOAT_ORACLE(double, f64, rint, floor, ceil)

/* ---------------------------------------------------------------------------------------------
 * Quantized forward: kernels/shifts_kernels.h:532-571 (nchwd_q), :574-624 (nhwdc_q);
 * quantized/shifts_quantized.cpp:18-100, :107-130.
 * Pure byte/int gather: shift = int_repr(w) - w_zero_point, fill = input zero point.
 * esize = 1 (quint8/qint8) or 4 (qint32); fill points at one element.
 * ------------------------------------------------------------------------------------------- */
void oracle_forward_q(const oracle_geom *g, int esize, const void *x_, const idx_t *wq,
                      idx_t w_zero_point, const void *fill, void *out_)
{
    const unsigned char *x = (const unsigned char *)x_;
    unsigned char *out = (unsigned char *)out_;
    const int nd = g->nd;
    const idx_t li = g->b[0], ri = g->b[1];
    const idx_t lj = nd < 2 ? 0 : g->b[2], rj = nd < 2 ? 1 : g->b[3];
    const idx_t lk = nd < 3 ? 0 : g->b[4], rk = nd < 3 ? 1 : g->b[5];
    /* results are independent of loop order; keep one nest */
    for (idx_t n = 0; n < g->N; ++n)
        for (idx_t c = 0; c < g->C; ++c)
            for (idx_t i = li; i < ri; ++i)
                for (idx_t j = lj; j < rj; ++j)
                    for (idx_t k = lk; k < rk; ++k) {
                        const idx_t oi = i - li, oj = nd > 1 ? j - lj : j, ok = nd > 2 ? k - lk : k;
                        const idx_t si = i - wq[c * nd] + w_zero_point;
                        const idx_t sj = nd > 1 ? j - wq[c * nd + 1] + w_zero_point : j;
                        const idx_t sk = nd > 2 ? k - wq[c * nd + 2] + w_zero_point : k;
                        idx_t off = gather_offset(g, g->pad, si, g->H, g->x_s[2], sj, g->W, g->x_s[3],
                                                  sk, g->D, g->x_s[4], c, g->x_s[1], 1);
                        unsigned char *o = out + (size_t)esize * (size_t)(n * g->o_s[0] + c * g->o_s[1] +
                                                 oi * g->o_s[2] + oj * g->o_s[3] + ok * g->o_s[4]);
                        if (off >= 0)
                            memcpy(o, x + (size_t)esize * (size_t)(n * g->x_s[0] + off), (size_t)esize);
                        else
                            memcpy(o, fill, (size_t)esize);
                    }
}

/* ---------------------------------------------------------------------------------------------
 * check_borders: ops/shifts.cpp:93-135.  sizes = full tensor sizes (rank nsizes), user = nD x 2
 * cut amounts or NULL.  Writes 6 ints [l_i,r_i,l_j,r_j,l_k,r_k] and the new sizes.
 * ------------------------------------------------------------------------------------------- */
void oracle_check_borders(const idx_t *sizes, int nsizes, const int32_t *user, int dim,
                          int32_t *std_b, idx_t *new_sizes)
{
    const int shift = ((dim + 1) == nsizes) ? 1 : 2;
    const int hdim = 3;
    const int _dim = dim < hdim ? dim : hdim;
    for (int i = 0; i < hdim; ++i) {
        std_b[i * 2] = 0;
        std_b[i * 2 + 1] = ((i + 1) > dim) ? 1 : (int32_t)sizes[i + shift];
    }
    if (user) {
        for (int i = 0; i < _dim; ++i) {
            const int32_t sz = (int32_t)sizes[i + shift];
            std_b[i * 2 + 1] -= user[i * 2 + 1];
            std_b[i * 2] = user[i * 2];
            if ((std_b[i * 2 + 1] - std_b[i * 2]) < 1) std_b[i * 2 + 1] = std_b[i * 2] + 1;
            if (std_b[i * 2] == sz) { std_b[i * 2] = sz - 1; std_b[i * 2 + 1] = std_b[i * 2] + 1; }
            if (std_b[i * 2 + 1] == 0) { std_b[i * 2] = 0; std_b[i * 2 + 1] = 1; }
            if (std_b[i * 2] < 0) std_b[i * 2] = 0;
            if (std_b[i * 2 + 1] > sz) std_b[i * 2 + 1] = sz;
        }
    }
    for (int i = 0; i < shift; ++i) new_sizes[i] = sizes[i];
    for (int i = 0; i < _dim; ++i) new_sizes[i + shift] = (idx_t)(std_b[i * 2 + 1] - std_b[i * 2]);
}

int oracle_abi_version(void) { return 1; }
