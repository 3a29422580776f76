// shiftnd_step.hpp -- what the one-step kernels (shiftnd_step.hip) and the walk kernels (shiftnd_walk.hip) share: the
// per-channel descriptor written by step_prep, the launch parameters, the weight-gradient reduction (step_reduce), the
// DPP wave sums, the packed 16-bit dot product and the scalar-cache weight loads.  Internal header (unnamed namespace:
// one private copy of the small kernels per translation unit).
#pragma once

#include <algorithm>
#include <type_traits>

#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"
#include "shiftnd_stage.hpp"

namespace shiftnd {

// knobs (shiftnd_set_tuning 32..): [0] backward, [1] gather forward (direct loads), [2] forwards through LDS: 0 = automatic, 1 = never, 2 = whenever eligible; [3] bit 1 / bit 2: one / two row groups per thread always, bit 3: 3-D forwards through LDS too, bit 4: no walk kernels, bit 5: round 3's walk kernels for fp64 too
extern thread_local int g_step_tune[5];   // [4]: planes per workgroup of the walk kernels (0 = all); defined in shiftnd_step.hip

// (the launch parameters are named types of the library: they cross translation units -- step_backward() hands its
//  StepParams to shiftnd_walk3.hip; the kernels and helpers below are private copies per translation unit)
struct ChanDesc {  // per channel, written by step_prep
    int cx0, cg0;  // plane maps (3-D; 0 for 2-D):  m0[p] = fold_index(p - cx0, S0, pad), g0[p] likewise with cg0
    int cx1, cg1;  // row maps:                     m1[p] = fold_index(p - cx1, S1, pad), g1[p] likewise with cg1
    int cx2, cg2;  // the column maps' canonical shifts (zeros padding: the column state is two compares, no table)
    int scat;      // 2-D sparse shift: the row shift clamped to [-S1, S1] (0 when S1 == 1)
    int pad_;
    double dw[3];  // fractions of prep_shift_backward per real dim, exactly as the compute type holds them
    double pad2_;
};

struct StepParams {
    const void *x;      // saved input
    const void *go;     // incoming gradient
    void *out;          // grad_x
    const void *w;
    double *partials;   // [total_steps][NDIFF]
    ChanDesc *desc;     // [C]
    int16_t *colx;      // [C][cpr][REC] column state of every chunk through the x column map
    int16_t *colg;      // ... through the grad column map
    int64_t x_plane;    // elements per (n, c) plane (2-D) / volume (3-D)
    int wkind, N, C, pad, nd;
    int S0, S1, S2;     // planes per volume (1 for 2-D), rows per plane, elements per row
    int cpr, R, spp;    // 16-byte chunks per row, rows per step, steps per plane
    int spv;            // steps per (n, c): S0 * spp
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_spp, d_spv, d_C, d_cpr, d_per0, d_per1, d_per2;
    // fused average-pool tail (2-D): `go` is the gradient of the POOLED output [N, C, P1, P2], window = stride = (K1, K2)
    int K1, K2, P1, P2;
    int64_t g_plane;    // elements per (n, c) plane of `go`
    FastDiv d_k1, d_k2;
    int K0, P0;         // 3-D pooled calls (walk_backward<..., POOL>): window and pooled size along dim0
    FastDiv d_k0;
    int walk_planes;    // walk_backward: planes a workgroup walks through (S0, or a part of the volume's depth)
    int crop;           // walk_backward<.., CROP> / walk_backward16<.., CROP>: `go` is the gradient of a window of the volume ...
    int wO0, wO1, wO2;  // ... of these sizes ...
    int wL0, wL1, wL2;  // ... that begins at this plane / row / column (with POOL: `go` is the pooled gradient of that window)
};

// the cropped walks (round 6): zeros padding, no pool, a window of at least 2 x 2 x 2 that begins at most two columns into the rows;
// 16-bit elements: window rows of an even number of elements (shiftnd_walk.hip), 4-byte elements: shiftnd_walk3.hip.  Knob 35 bit 11
// keeps the one-step crop_backward3.
inline bool walk_crop_window_ok(const Geometry &g, bool pooled = false) {
    if (g.nd != 3 || g.pad != 0 || (g.K[0] > 0) != pooled || (g_step_tune[3] & 2048)) return false;
    bool cropped = false;
    for (int d = 0; d < 3; ++d) {
        if (g.O[d] < 2 || g.L[d] < 0 || g.L[d] + g.O[d] > g.S[d]) return false;
        cropped = cropped || g.O[d] != g.S[d] || g.L[d] != 0;
    }
    return cropped && g.L[2] <= 2;
}

// the second half of step_backward()'s launch for 3-D problems (shiftnd_walk3.hip)
int walk3_backward_launch(StepParams &p, const Geometry &g, int dtype, int cpr, void *gw, hipStream_t st);

// One copy of step_prep / step_reduce in the library: shiftnd_step.hip instantiates them and defines these launchers; the walk
// and row-span translation units call them (T::kDtype picks the element type).
void launch_step_prep(int dtype, bool active, const StepParams &p, hipStream_t st);
void launch_step_reduce(int dtype, int nd, const StepParams &p, void *grad_w, hipStream_t st);

namespace {

template <int E> struct RecSize { static constexpr int N = (E + 3 <= 8) ? 8 : 16; };  // int16 entries per record

__device__ __forceinline__ int row_map(int p, int cs, int len, int pad) { return len == 1 ? 0 : fold_index(p - cs, len, pad); }

// ---------------------------------------------------------------------------------------------------------------------
// step_prep: one workgroup per channel -- the weight preparation of the reference (shifts_cuda.cu:168-199) plus the
// channel's column maps in the form the step kernels read them
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void step_prep(const StepParams p, const bool ACTIVE) {   // (ACTIVE: a launch argument -- one kernel per dtype)
    using S = typename T::S;
    using CT = typename T::C;
    constexpr int E = 16 / sizeof(S);
    constexpr int REC = RecSize<E>::N;
    const int c = blockIdx.x;
    // real dim r of an nd-dim problem is normalised dim r + 3 - nd: (plane,) row, inner
    const int lead = 3 - p.nd;
    int64_t sh[3] = {0, 0, 0};
    CT dw[3] = {CT(0), CT(0), CT(0)};
    for (int r = 0; r < p.nd; ++r) {
        const CT wv = load_weight<CT>(p.w, p.wkind, static_cast<int64_t>(c) * p.nd + r);
        prep_shift_backward<CT>(wv, ACTIVE, sh[r + lead], dw[r]);
    }
    // build_maps: x map with sign -1 -> canon_shift(sh); grad map with sign +1 (sparse) -> canon_shift(-sh), -1 (active)
    const int cx0 = canon_shift(sh[0], p.S0, p.pad, p.d_per0), cx1 = canon_shift(sh[1], p.S1, p.pad, p.d_per1),
              cx2 = canon_shift(sh[2], p.S2, p.pad, p.d_per2);
    const int cg0 = canon_shift(ACTIVE ? sh[0] : -sh[0], p.S0, p.pad, p.d_per0),
              cg1 = canon_shift(ACTIVE ? sh[1] : -sh[1], p.S1, p.pad, p.d_per1),
              cg2 = canon_shift(ACTIVE ? sh[2] : -sh[2], p.S2, p.pad, p.d_per2);
    if (threadIdx.x == 0) {
        ChanDesc d;
        d.cx0 = cx0;
        d.cg0 = cg0;
        d.cx1 = cx1;
        d.cg1 = cg1;
        d.cx2 = cx2;
        d.cg2 = cg2;
        d.scat = p.S1 == 1 ? 0 : static_cast<int>(sh[1] > p.S1 ? p.S1 : (sh[1] < -p.S1 ? -p.S1 : sh[1]));
        d.pad_ = 0;
        d.dw[0] = static_cast<double>(dw[0]);
        d.dw[1] = static_cast<double>(dw[1]);
        d.dw[2] = static_cast<double>(dw[2]);
        d.pad2_ = 0.0;
        p.desc[c] = d;
    }
    if (p.pad == 0 || !p.colx) return;  // zeros padding (and the round-4 walk): the kernels fold the column state themselves
    for (int j = threadIdx.x; j < p.cpr; j += kThreads) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int cs = which ? cg2 : cx2;
            int cm[E + 1];
            int base = 0;
            bool found = false, affine = true;
#pragma unroll
            for (int e = 0; e <= E; ++e) {
                cm[e] = row_map(j * E + e, cs, p.S2, p.pad);
                if (!found && cm[e] >= 0) {
                    base = cm[e] - e;
                    found = true;
                }
            }
#pragma unroll
            for (int e = 0; e <= E; ++e) affine = affine && (cm[e] < 0 || cm[e] == base + e);
            int16_t *rec = (which ? p.colg : p.colx) + (static_cast<size_t>(c) * p.cpr + j) * REC;
#pragma unroll
            for (int e = 0; e <= E; ++e) rec[e] = static_cast<int16_t>(cm[e]);
            rec[E + 1] = static_cast<int16_t>(base);
            rec[E + 2] = affine ? 1 : 0;
        }
    }
}

template <int E> __device__ __forceinline__ ColState<E> load_colstate(const int16_t *rec) {
    constexpr int REC = RecSize<E>::N;
    int16_t r[REC];
    const Chunk<int16_t, 8> a = load_chunk<int16_t, 8>(rec);
    __builtin_memcpy(r, a.e, 16);
    if constexpr (REC == 16) {
        const Chunk<int16_t, 8> b = load_chunk<int16_t, 8>(rec + 8);
        __builtin_memcpy(r + 8, b.e, 16);
    }
    ColState<E> c;
#pragma unroll
    for (int e = 0; e <= E; ++e) c.cm[e] = r[e];
    c.base = r[E + 1];
    c.affine = r[E + 2] != 0;
    return c;
}

// fixed-order sum over the 64 lanes of a wave, result in lane 63.  fp32: six v_add_f32 with DPP operands (row shifts,
// then the row broadcasts of GFX9); fp64: the shuffle tree (fp64 tensors are rare, the DPP form moves 32 bits)
__device__ __forceinline__ float wave_total(float v) {
#define SHIFTND_DPP_ADD(CTRL, ROWMASK) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, true))
    SHIFTND_DPP_ADD(0x111, 0xf);  // row_shr:1
    SHIFTND_DPP_ADD(0x112, 0xf);  // row_shr:2
    SHIFTND_DPP_ADD(0x114, 0xf);  // row_shr:4
    SHIFTND_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds its row's sum
    SHIFTND_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3 add the row below
    SHIFTND_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3 add lane 31
#undef SHIFTND_DPP_ADD
    return v;
}
__device__ __forceinline__ double wave_total(double v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        v += (static_cast<int>(threadIdx.x & 63) >= off) ? o : 0.0;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// step_backward.  PAD is a template parameter: the row maps are folded per staged piece and per output row, and a
// run-time padding switch there costs more scalar-unit time than a one-step workgroup has (one scalar unit per CU;
// shiftnd_common.hpp fold_index is 2 - 5 VALU instructions once the mode is known).
// ---------------------------------------------------------------------------------------------------------------------
// PAD as a template parameter takes the values 0 .. 3, and 3 stands for BOTH mirroring modes: reflect (3) and symmetric (4) differ
// by a constant -- k = pad - 3 -- in each of their two folds,
//     idx < 0:        -idx      / -idx - 1         = max(idx, -k - idx)
//     idx > len - 1:  2 (len - 1) - idx / 2 len - 1 - idx = min(t, 2 (len - 1) + k - t)
// two VALU instructions each with k in a scalar register: the same count as the compile-time forms, one instantiation instead of two.
// `rt` is the kernel's padding argument (p.pad); it only matters for PAD = 3.  (A run-time mode for all four non-zeros paddings was
// measured on the row-span kernels: + 13 .. 30 % on their forwards, + 5 .. 20 % on the backwards -- ten VALU instructions per fold
// against two to five; the flat-stream kernels, whose per-element path hides it, merge the three wrapping modes: shiftnd_flat.hip.)
constexpr int kPadMirror = 3;
constexpr int pad_template(int pad) { return pad >= kPadMirror ? kPadMirror : pad; }

__device__ __forceinline__ int fold_mirror(int idx, int len, int k) {
    const int t = max(idx, -k - idx);
    return min(t, 2 * (len - 1) + k - t);
}

template <int PAD> __device__ __forceinline__ int row_map_t(int p, int cs, int len, int rt = kPadMirror) {
    if constexpr (PAD == kPadMirror) return len == 1 ? 0 : fold_mirror(p - cs, len, rt - kPadMirror);
    else return len == 1 ? 0 : fold_index(p - cs, len, PAD);
}

// canon_shift (shiftnd_common.hpp) for |s| < 2^30 and a compile-time padding mode, all in 32 bits
template <int PAD> __device__ __forceinline__ int canon_shift32(int s, int len, const FastDiv &dper, int rt = kPadMirror) {
    if (len <= 1) return 0;
    if constexpr (PAD <= 1) {
        return s < -len - 1 ? -len - 1 : (s > len + 1 ? len + 1 : s);
    } else {
        const int period = PAD == 2 ? len : 2 * (len - 1) + 2 * (rt - kPadMirror);   // reflect 2 (len - 1), symmetric 2 len
        const uint32_t a = static_cast<uint32_t>(s < 0 ? -s : s);
        const uint32_t m = a - fdiv(a, dper) * static_cast<uint32_t>(period);
        return static_cast<int>((s < 0 && m != 0) ? static_cast<uint32_t>(period) - m : m);
    }
}

// acc + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): the products of two 16-bit
// values are exact in fp32 and nothing has to be widened first
template <typename T> __device__ __forceinline__ float dot2_packed(uint32_t a, uint32_t b, float c) {
    if constexpr (T::kDtype == SHIFTND_BF16) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    } else {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
}

// the two weights of channel c (row dim, inner dim) widened to the compute type, through the scalar cache
template <typename CT> __device__ __forceinline__ void load_weights2(const void *w, int wkind, int c, CT &wr, CT &wc) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(w);
    if (wkind == SHIFTND_F64) {
        const __attribute__((address_space(4))) double *q = reinterpret_cast<const __attribute__((address_space(4))) double *>(base) + static_cast<int64_t>(c) * 2;
        wr = static_cast<CT>(q[0]);
        wc = static_cast<CT>(q[1]);
    } else if (wkind == SHIFTND_F16 || wkind == SHIFTND_BF16) {
        // two halfwords at byte offset 4 c of a base that is 2-byte aligned at least: the two aligned dwords that hold them
        const uintptr_t at = base + static_cast<uintptr_t>(c) * 4;
        const __attribute__((address_space(4))) uint32_t *q = reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(at & ~static_cast<uintptr_t>(3));
        const uint32_t word = (at & 2) ? ((q[0] >> 16) | (q[1] << 16)) : q[0];
        const uint16_t lo = static_cast<uint16_t>(word), hi = static_cast<uint16_t>(word >> 16);
        if (wkind == SHIFTND_F16) {
            wr = static_cast<CT>(__builtin_bit_cast(_Float16, lo));
            wc = static_cast<CT>(__builtin_bit_cast(_Float16, hi));
        } else {
            wr = static_cast<CT>(__builtin_bit_cast(float, static_cast<uint32_t>(lo) << 16));
            wc = static_cast<CT>(__builtin_bit_cast(float, static_cast<uint32_t>(hi) << 16));
        }
    } else {
        const __attribute__((address_space(4))) float *q = reinterpret_cast<const __attribute__((address_space(4))) float *>(base) + static_cast<int64_t>(c) * 2;
        wr = static_cast<CT>(q[0]);
        wc = static_cast<CT>(q[1]);
    }
}

// canonical shift of an integral shift held in the compute type: 32-bit arithmetic below 2^30, the 64-bit form beyond
template <int PAD, typename CT> __device__ __forceinline__ int canon_of(CT r, int len, const FastDiv &dper, int rt = kPadMirror) {
    if (r > CT(-1073741824) && r < CT(1073741824)) return canon_shift32<PAD>(static_cast<int>(r), len, dper, rt);
    return canon_shift(static_cast<int64_t>(r), len, PAD == kPadMirror ? rt : PAD, dper);
}

// the two canonical shifts of channel c for the gather kernels (sparse shift: round half to even; quantized: int_repr minus
// zero point, kernels/shifts_kernels.h:553-555), every weight dtype through the scalar cache
template <int PAD>
__device__ __forceinline__ void channel_shifts2(const void *w, int wkind, int64_t wzp, int c, int S1, int S2, const FastDiv &d1,
                                                const FastDiv &d2, int &cs1, int &cs2, int rt = kPadMirror) {
    if (wkind <= SHIFTND_BF16) {
        if (wkind == SHIFTND_F64) {
            double wr, wc;
            load_weights2<double>(w, wkind, c, wr, wc);
            cs1 = canon_of<PAD, double>(rint(wr), S1, d1, rt);
            cs2 = canon_of<PAD, double>(rint(wc), S2, d2, rt);
        } else {
            float wr, wc;
            load_weights2<float>(w, wkind, c, wr, wc);
            cs1 = canon_of<PAD, float>(rintf(wr), S1, d1, rt);
            cs2 = canon_of<PAD, float>(rintf(wc), S2, d2, rt);
        }
    } else {
        const uintptr_t base = reinterpret_cast<uintptr_t>(w);
        int64_t r1, r2;
        if (wkind == SHIFTND_I32) {
            const __attribute__((address_space(4))) int32_t *q = reinterpret_cast<const __attribute__((address_space(4))) int32_t *>(base) + static_cast<int64_t>(c) * 2;
            r1 = q[0];
            r2 = q[1];
        } else {  // two bytes at byte offset 2 c: the aligned dword that holds them
            const uintptr_t at = base + static_cast<uintptr_t>(c) * 2;
            const uint32_t word = *reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(at & ~static_cast<uintptr_t>(3));
            const uint32_t pair = word >> ((at & 2) * 8);
            if (wkind == SHIFTND_I8) {
                r1 = static_cast<int8_t>(pair & 0xff);
                r2 = static_cast<int8_t>((pair >> 8) & 0xff);
            } else {
                r1 = pair & 0xff;
                r2 = (pair >> 8) & 0xff;
            }
        }
        r1 -= wzp;
        r2 -= wzp;
        const bool small = r1 > -1073741824 && r1 < 1073741824 && r2 > -1073741824 && r2 < 1073741824;
        cs1 = small ? canon_shift32<PAD>(static_cast<int>(r1), S1, d1, rt) : canon_shift(r1, S1, PAD == kPadMirror ? rt : PAD, d1);
        cs2 = small ? canon_shift32<PAD>(static_cast<int>(r2), S2, d2, rt) : canon_shift(r2, S2, PAD == kPadMirror ? rt : PAD, d2);
    }
    cs1 = __builtin_amdgcn_readfirstlane(cs1);
    cs2 = __builtin_amdgcn_readfirstlane(cs2);
}

struct GatherParams {  // 2-D problems: rows x inner, one weight per dim
    const void *x;
    void *out;
    const void *w;
    int64_t wzp;
    uint64_t fill;
    int64_t x_plane, o_plane;
    int wkind, C, pad;   // pad: the padding mode (the kPadMirror instantiations: reflect or symmetric)
    int S1, S2, O1, O2, L1, L2;
    int cpr, R, spp;
    int xppr;   // 16-byte pieces per source row (the small-element kernel)
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_spp, d_C, d_cpr, d_per1, d_per2;
    // 3-D (step_gather_forward<..., 3>: float weights): planes of the volume, steps per (n, c) = O0 * spp
    int S0, O0, L0, spv;
    FastDiv d_spv, d_per0;
};

// the nd weights of channel c in normalised order (plane, row, inner; leading dims 0), widened, through the scalar cache
template <typename CT> __device__ __forceinline__ void load_weights_nd(const void *w, int wkind, int c, int nd, CT (&out)[3]) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(w);
    CT v[3] = {CT(0), CT(0), CT(0)};
    if (wkind == SHIFTND_F64) {
        const __attribute__((address_space(4))) double *q = reinterpret_cast<const __attribute__((address_space(4))) double *>(base) + static_cast<int64_t>(c) * nd;
        for (int r = 0; r < 3; ++r) if (r < nd) v[r] = static_cast<CT>(q[r]);
    } else if (wkind == SHIFTND_F16 || wkind == SHIFTND_BF16) {
        // nd halfwords at byte offset 2 nd c: the two aligned dwords that hold them
        const uintptr_t at = base + static_cast<uintptr_t>(c) * nd * 2;
        const __attribute__((address_space(4))) uint32_t *q = reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(at & ~static_cast<uintptr_t>(3));
        const uint64_t bits = (static_cast<uint64_t>(q[1]) << 32 | q[0]) >> ((at & 2) * 8);
        for (int r = 0; r < 3; ++r) {
            if (r >= nd) break;
            const uint16_t h = static_cast<uint16_t>(bits >> (16 * r));
            v[r] = wkind == SHIFTND_F16 ? static_cast<CT>(__builtin_bit_cast(_Float16, h))
                                        : static_cast<CT>(__builtin_bit_cast(float, static_cast<uint32_t>(h) << 16));
        }
    } else {
        const __attribute__((address_space(4))) float *q = reinterpret_cast<const __attribute__((address_space(4))) float *>(base) + static_cast<int64_t>(c) * nd;
        for (int r = 0; r < 3; ++r) if (r < nd) v[r] = static_cast<CT>(q[r]);
    }
    const int lead = 3 - nd;
    out[0] = out[1] = out[2] = CT(0);
    for (int r = 0; r < 3; ++r) if (r < nd) out[r + lead] = v[r];
}

struct FwdParams {
    const void *x;
    void *out;
    const void *w;
    uint64_t fill;
    int64_t x_plane, o_plane;  // elements per (n, c) plane (2-D) / volume (3-D)
    int wkind, C, nd;
    int pad;   // (the round-4 walk kernels: the padding mode is a run-time value there)
    int S0, S1, S2, O0, O1, O2, L0, L1, L2;
    int cpr, xppr, R, spp;   // output chunks per row, source pieces per row, rows per step, steps per plane
    int spv;                 // steps per (n, c): O0 * spp
    uint32_t total_steps, steps_per_xcd;
    FastDiv d_spp, d_spv, d_C, d_cpr, d_xppr, d_per0, d_per1, d_per2;
    // walk_forward<..., POOL>: average pool (K0, K1, 2) behind the shift; `out` is the pooled tensor [N, C, P0, P1, P2]
    int K0, K1, P1, P2;
    int64_t p_plane;   // pooled elements per (n, c)
    FastDiv d_k0;
};

// column state of E + 1 consecutive map entries starting at coordinate j0, folded arithmetically
template <int E, int PAD> __device__ __forceinline__ ColState<E> fold_colstate(int j0, int cs, int len, int rt = kPadMirror) {
    ColState<E> c;
    c.base = 0;
    bool found = false;
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        c.cm[e] = row_map_t<PAD>(j0 + e, cs, len, rt);
        if (!found && c.cm[e] >= 0) {
            c.base = c.cm[e] - e;
            found = true;
        }
    }
    c.affine = true;
#pragma unroll
    for (int e = 0; e <= E; ++e) c.affine = c.affine && (c.cm[e] < 0 || c.cm[e] == c.base + e);
    return c;
}

// grad_w[c][0..nd-1] = blend(sum over the steps of channel c, in a fixed order)
template <typename T, int ND>
__device__ __forceinline__ void step_reduce_nd(const StepParams &p, typename T::S *__restrict__ grad_w, double *scratch) {
    constexpr int NDIFF = WDiff<ND>::N;
    const int c = blockIdx.x;
    const uint32_t per_channel = static_cast<uint32_t>(p.N) * static_cast<uint32_t>(p.spv);
    double acc[NDIFF];
#pragma unroll
    for (int i = 0; i < NDIFF; ++i) acc[i] = 0.0;
    for (uint32_t i = threadIdx.x; i < per_channel; i += kThreads) {
        const uint32_t n = fdiv(i, p.d_spv);
        const uint32_t st = i - n * static_cast<uint32_t>(p.spv);
        const double *q = p.partials + ((static_cast<size_t>(n) * p.C + c) * p.spv + st) * NDIFF;
#pragma unroll
        for (int k = 0; k < NDIFF; ++k) acc[k] += q[k];
    }
    double dsum[NDIFF];
#pragma unroll
    for (int k = 0; k < NDIFF; ++k) dsum[k] = block_sum(acc[k], scratch);
    if (threadIdx.x == 0) {
        const double dwd[3] = {p.desc[c].dw[0], p.desc[c].dw[1], p.desc[c].dw[2]};
        double out[3] = {0.0, 0.0, 0.0};
        blend_diffs<ND>(dsum, dwd, out);
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            if constexpr (sizeof(typename T::S) == 8) grad_w[c * ND + s] = out[s];
            else grad_w[c * ND + s] = narrow<T>(static_cast<float>(out[s]));
        }
    }
}

// (one kernel per dtype: the number of dims is a launch argument that picks the body)
template <typename T>
__global__ __launch_bounds__(kThreads) void step_reduce(const StepParams p, typename T::S *__restrict__ grad_w, const int nd) {
    __shared__ double scratch[kThreads / 64];
    if (nd == 1) step_reduce_nd<T, 1>(p, grad_w, scratch);
    else if (nd == 2) step_reduce_nd<T, 2>(p, grad_w, scratch);
    else step_reduce_nd<T, 3>(p, grad_w, scratch);
}


// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
bool dense(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

// ---- the launch plan of step_backward and of the round-3 walk kernels (host) ---------------------------------------------
struct StepLayout {
    int cpr, R, U, spp, spv, rec, ndiff;
    uint64_t total_steps;
    size_t off_desc, off_colx, off_colg, bytes;
};

// row groups per thread (knob 35 bit 1 = 2: always one): two for 16-bit data, where the kernel is
// bound by instruction issue (same box, one vs two: fp16 C512 224x224 reflect 1.80 -> 1.67 ms, interpolating 1.91 -> 1.79,
// bf16 N128 C256 56x56 0.126 -> 0.109; zeros padding 1.61 vs 1.62); one for 4- / 8-byte elements (fp32 sparse 1.58 vs 1.62,
// interpolating 1.63 vs 1.67 ms: the tighter sweep front wins), 3-D and pooled calls
int step_row_groups(const Geometry &g, int es) {
    if (g.nd != 2 || g.K[0] > 0) return 1;
    if (g_step_tune[3] & 2) return 1;
    return es == 2 ? 2 : 1;   // (two row groups are built for 16-bit data only)
}

// force_u: row groups per thread (0: by dtype and knob 35).  The WORKSPACE is planned with one (the most steps), so that its size
// does not depend on the thread-local knobs of whoever asks; a run lays its regions out with its own U inside that.
StepLayout step_layout(const Geometry &g, int es, int force_u = 0) {
    StepLayout L{};
    const int E = 16 / es;
    L.cpr = static_cast<int>(g.S[2] * es / 16);
    if (L.cpr < 1) L.cpr = 1;
    L.R = kThreads / L.cpr < 1 ? 1 : kThreads / L.cpr;
    if (L.R > g.S[1]) L.R = static_cast<int>(g.S[1] > 0 ? g.S[1] : 1);
    L.U = force_u > 0 ? force_u : step_row_groups(g, es);
    if (L.U * L.R > g.S[1] && L.R >= g.S[1]) L.U = 1;  // (one row group already covers the plane)
    L.spp = static_cast<int>((g.S[1] + L.U * L.R - 1) / (L.U * L.R));
    L.spv = static_cast<int>(g.S[0]) * L.spp;
    L.rec = (E + 3 <= 8) ? 8 : 16;
    L.ndiff = g.nd == 3 ? 8 : 2;
    L.total_steps = static_cast<uint64_t>(g.N) * g.C * L.spv;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    L.off_desc = up(L.total_steps * L.ndiff * sizeof(double));
    L.off_colx = L.off_desc + up(static_cast<size_t>(g.C) * sizeof(ChanDesc));
    L.off_colg = L.off_colx + up(static_cast<size_t>(g.C) * L.cpr * L.rec * sizeof(int16_t));
    L.bytes = L.off_colg + up(static_cast<size_t>(g.C) * L.cpr * L.rec * sizeof(int16_t));
    return L;
}

size_t step_lds_bytes(const StepLayout &L, int nd, bool active) {
    const int np = nd == 3 ? 2 : 1;
    const int RT = L.U * L.R;
    const int slots = np * (RT + 1) + RT + (active ? np * (RT + 1) : (nd == 3 ? RT : 0));
    return 64 + ((static_cast<size_t>(slots) * L.cpr * 16 + 63) & ~static_cast<size_t>(63)) + (kThreads / 64) * L.ndiff * sizeof(double);
}

}  // namespace
}  // namespace shiftnd
