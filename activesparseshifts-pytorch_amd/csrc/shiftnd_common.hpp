// shiftnd_common.hpp -- shared device/host helpers of the MI355X shiftnd kernels (gfx950 only).
//
// Semantics follow the reference's per-element math (torchshifts/csrc/ops/kernels/shifts_kernels.h,
// interpolation.h); the structure (per-plane workgroups, LDS index maps, 16-byte row chunks,
// fp64 two-stage weight-gradient reduction) is native to this library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shiftnd_hip.h"

namespace shiftnd {

constexpr int kThreads = 256;        // workgroup size of every kernel (4 waves of 64)
constexpr int kMaxMapEntries = 12288; // LDS budget for index maps (48 KiB of int32)

// ---------------------------------------------------------------------------------------------
// Element types.  Storage type S (what lives in HBM), compute type C (what interpolation uses).
// ---------------------------------------------------------------------------------------------
struct f32_t { using S = float;    using C = float;  static constexpr int kDtype = SHIFTND_F32; };
struct f64_t { using S = double;   using C = double; static constexpr int kDtype = SHIFTND_F64; };
struct f16_t { using S = _Float16; using C = float;  static constexpr int kDtype = SHIFTND_F16; };
struct bf16_t { using S = __bf16;  using C = float;  static constexpr int kDtype = SHIFTND_BF16; };

template <typename T> __device__ __forceinline__ typename T::C widen(typename T::S v) {
    return static_cast<typename T::C>(v);
}
// one rounding (RNE) from the compute type to the storage type
template <typename T> __device__ __forceinline__ typename T::S narrow(typename T::C v) {
    return static_cast<typename T::S>(v);
}

// ---------------------------------------------------------------------------------------------
// Division by a launch-invariant 32-bit divisor (n < 2^31): q = (umulhi(n, mul) + n) >> shift.
// ---------------------------------------------------------------------------------------------
struct FastDiv {
    uint32_t d, mul, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    uint32_t s = 0;
    while (s < 32 && (1ull << s) < f.d) ++s;
    f.shift = s;
    f.mul = static_cast<uint32_t>(((1ull << 32) * ((1ull << s) - f.d)) / f.d + 1);
    return f;
}
__host__ __device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    const uint32_t hi = static_cast<uint32_t>((static_cast<uint64_t>(n) * f.mul) >> 32);  // v_mul_hi_u32
    return (hi + n) >> f.shift;
}

// ---------------------------------------------------------------------------------------------
// Padding index map: infer_index (shifts_kernels.h:10-29) + the `tidx >= 0` validity test of
// get_shifted_value (:40-41): returns the source index in [0, len) or -1 for "use the fill value".
// 64-bit so that arbitrarily large shifts (multi-wrap) stay exact.  len == 1 is handled by the
// callers (size-1 dims ignore the shift, :40).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t pmod(int64_t a, int64_t b) { return (b + (a % b)) % b; }

__host__ __device__ inline int64_t pad_index(int64_t idx, int64_t len, int pad) {
    int64_t r;
    switch (pad) {
    case 1:  // border
        r = idx < 0 ? 0 : (idx > len - 1 ? len - 1 : idx);
        break;
    case 2:  // periodic
        r = pmod(idx, len);
        break;
    case 3: {  // reflect, period 2*(len-1)
        const int64_t neg = idx < 0 ? 1 : 0;
        const int64_t a = idx < 0 ? -idx : idx;
        const bool odd = ((neg + (a - neg) / (len - 1)) & 1) != 0;
        const int64_t m = pmod(idx, len - 1);
        r = odd ? (len - 1 - m) : m;
        break;
    }
    case 4: {  // symmetric, period 2*len
        const int64_t neg = idx < 0 ? 1 : 0;
        const int64_t a = idx < 0 ? -idx : idx;
        const bool odd = ((neg + (a - neg) / len) & 1) != 0;
        const int64_t m = pmod(idx, len);
        r = odd ? (len - 1 - m) : m;
        break;
    }
    default:  // zeros: idx > len-1 -> fill; negatives are rejected by the >= 0 test
        r = (idx > len - 1) ? -1 : idx;
        break;
    }
    return r < 0 ? -1 : r;
}

// ---------------------------------------------------------------------------------------------
// Cheap per-element form of the same map, used by the sweep kernels (no tables):
//   canon_shift() reduces a shift to a canonical representative with the same index map
//   (zeros/border: clamped to [-len-1, len+1]; periodic: mod len; reflect: mod 2(len-1); symmetric:
//   mod 2len), after which idx = p - shift lies within one period of [0, len) for every p in
//   [0, len] and fold_index() needs at most two conditional folds.
//   fold_index(p - canon_shift(s), len, pad) == pad_index(p - s, len, pad) for all p in [0, len];
//   checked exhaustively on the host by tests/test_host_logic.py through shiftnd_debug_map().
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int map_period(int len, int pad) {
    switch (pad) {
    case 2: return len;
    case 3: return 2 * (len - 1);
    case 4: return 2 * len;
    default: return 0;  // zeros / border: no period, shifts are clamped
    }
}

// `dper` divides by map_period(len, pad) (host-built; ignored when the period is 0).
// No hardware integer division is used: hipcc (ROCm 7.2) was seen to miscompile a 32-bit signed
// remainder by a launch-uniform divisor in this function, and the multiply-shift form is cheaper.
__host__ __device__ inline int canon_shift(int64_t s, int len, int pad, const FastDiv &dper) {
    if (len <= 1) return 0;
    const int period = map_period(len, pad);
    if (period == 0) return static_cast<int>(s < -len - 1 ? -len - 1 : (s > len + 1 ? len + 1 : s));  // p ranges over [0, len]
    if (s >= -0x40000000LL && s <= 0x40000000LL) {  // every sane shift: 32-bit multiply-shift remainder
        const uint32_t a = static_cast<uint32_t>(s < 0 ? -s : s);
        const uint32_t m = a - fdiv(a, dper) * static_cast<uint32_t>(period);
        return static_cast<int>((s < 0 && m != 0) ? static_cast<uint32_t>(period) - m : m);
    }
    return static_cast<int>(pmod(s, period));
}

__host__ __device__ __forceinline__ int fold_index(int idx, int len, int pad) {
    switch (pad) {
    case 1: return idx < 0 ? 0 : (idx > len - 1 ? len - 1 : idx);
    case 2:
        idx += idx < 0 ? len : 0;
        return idx >= len ? idx - len : idx;
    case 3:
        idx = idx < 0 ? -idx : idx;
        return idx > len - 1 ? 2 * (len - 1) - idx : idx;
    case 4:
        idx = idx < 0 ? -idx - 1 : idx;
        return idx >= len ? 2 * len - 1 - idx : idx;
    default: return (idx < 0 || idx >= len) ? -1 : idx;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS index maps of the per-channel kernels (one workgroup = one channel = one shift per dim):
// map_d[p] = pad_index(p + sign*shift_d) for p in [0, size_d] (size_d + 1 entries: the interpolating
// kernels also read coordinate p + 1), the three normalised dims back to back.
// ---------------------------------------------------------------------------------------------
// Entry p of dim d = pad_index(p + sign * sh[d]) = fold_index(p - canon_shift(-sign * sh[d])) (the identity above):
// 32-bit selects per entry instead of pad_index's 64-bit divisions, and nothing is indexed dynamically (a dynamic index
// into the kernel-argument arrays is a vector load from memory per entry).  `dper[d]` divides by
// map_period(size[d], pad).  The prologue of every per-channel workgroup runs this: it is on the critical path of
// workgroups that live for a few tens of microseconds.
__device__ __forceinline__ void build_maps(int *maps, const int size[3], const int64_t sh[3], int sign, int pad,
                                           const FastDiv dper[3]) {
    const int l0 = size[0], l1 = size[1], l2 = size[2];
    const int c0 = canon_shift(sign < 0 ? sh[0] : -sh[0], l0, pad, dper[0]);
    const int c1 = canon_shift(sign < 0 ? sh[1] : -sh[1], l1, pad, dper[1]);
    const int c2 = canon_shift(sign < 0 ? sh[2] : -sh[2], l2, pad, dper[2]);
    const int n0 = l0 + 1, n1 = l1 + 1, n2 = l2 + 1;
    for (int t = threadIdx.x; t < n0 + n1 + n2; t += kThreads) {
        const int d = t < n0 ? 0 : (t < n0 + n1 ? 1 : 2);
        const int p = t - (d == 0 ? 0 : (d == 1 ? n0 : n0 + n1));
        const int len = d == 0 ? l0 : (d == 1 ? l1 : l2);
        const int cs = d == 0 ? c0 : (d == 1 ? c1 : c2);
        maps[t] = (len == 1) ? 0 : fold_index(p - cs, len, pad);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-channel shift preparation (cpu/shifts_cpu.cpp:223-224 forward, :242-244 backward).
// Rounding is half-to-even (torch::round on the CPU path), i.e. rint in the default mode.
// ---------------------------------------------------------------------------------------------
template <typename CT> __device__ __forceinline__ CT c_floor(CT v);
template <> __device__ __forceinline__ float c_floor<float>(float v) { return floorf(v); }
template <> __device__ __forceinline__ double c_floor<double>(double v) { return floor(v); }
template <typename CT> __device__ __forceinline__ CT c_ceil(CT v);
template <> __device__ __forceinline__ float c_ceil<float>(float v) { return ceilf(v); }
template <> __device__ __forceinline__ double c_ceil<double>(double v) { return ceil(v); }
template <typename CT> __device__ __forceinline__ CT c_rint(CT v);
template <> __device__ __forceinline__ float c_rint<float>(float v) { return rintf(v); }
template <> __device__ __forceinline__ double c_rint<double>(double v) { return rint(v); }

template <typename CT>
__device__ __forceinline__ void prep_shift_forward(CT w, bool active, int64_t &iw, CT &dw) {
    const CT r = active ? c_floor<CT>(w) : c_rint<CT>(w);
    iw = static_cast<int64_t>(r);
    dw = active ? (w - static_cast<CT>(iw)) : CT(0);
}
template <typename CT>
__device__ __forceinline__ void prep_shift_backward(CT w, bool active, int64_t &iw, CT &dw) {
    dw = active ? (w - c_floor<CT>(w)) : ((w > CT(0)) ? (w - c_floor<CT>(w)) : (c_ceil<CT>(w) - w));
    const CT r = active ? (w - dw) : c_rint<CT>(w);
    iw = static_cast<int64_t>(r);
}

// ---------------------------------------------------------------------------------------------
// Interpolation (kernels/interpolation.h:3-61).  The library is compiled with -ffp-contract=off so
// v1*(1-x)+v2*x is two multiplies and one add, as in the reference's x86-64 CPU build.
// Corner order (shifts_kernels.h:58-103): bit0 = +1 along H, bit1 = +1 along W, bit2 = +1 along D.
// ---------------------------------------------------------------------------------------------
// fused multiply-add, for the weight-gradient sums and the 16-bit interpolation (the library is built with -ffp-contract=off because the
// interpolation must round like the reference's mul + mul + add; the sums carry a 1e-5 tolerance and are exact
// either way on exactly representable data)
__device__ __forceinline__ float fma_ct(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_ct(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename CT> __device__ __forceinline__ CT lerp1(CT v1, CT v2, CT x) { return v1 * (CT(1) - x) + v2 * x; }

template <int ND, typename CT> __device__ __forceinline__ CT interp_nd(const CT *v, const CT *d) {
    if constexpr (ND == 1) {
        return lerp1(v[0], v[1], d[0]);
    } else if constexpr (ND == 2) {
        return lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]);
    } else {
        const CT a = lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]);
        const CT b = lerp1(lerp1(v[4], v[5], d[0]), lerp1(v[6], v[7], d[0]), d[1]);
        return lerp1(a, b, d[2]);
    }
}

// Interpolation as the kernels call it.  fp32 / fp64 tensors: the reference's expression, bit for bit.  16-bit
// tensors have no executable reference arithmetic (fp32 math on widened inputs, one rounding to
// the 16-bit type, compared within 1 ulp of that type), so each lerp is v1*(1-x) folded into one multiply and one
// fused multiply-add: a third fewer instructions in kernels that are instruction-bound for 16-bit data, and not less
// accurate.  Every kernel family uses this one definition, so they agree bit for bit with each other.
template <typename CT> __device__ __forceinline__ CT lerp1_fused(CT v1, CT v2, CT x) { return fma_ct(v2, x, v1 * (CT(1) - x)); }
template <int ND, typename CT> __device__ __forceinline__ CT interp_nd_fused(const CT *v, const CT *d) {
    if constexpr (ND == 1) {
        return lerp1_fused(v[0], v[1], d[0]);
    } else if constexpr (ND == 2) {
        return lerp1_fused(lerp1_fused(v[0], v[1], d[0]), lerp1_fused(v[2], v[3], d[0]), d[1]);
    } else {
        const CT a = lerp1_fused(lerp1_fused(v[0], v[1], d[0]), lerp1_fused(v[2], v[3], d[0]), d[1]);
        const CT b = lerp1_fused(lerp1_fused(v[4], v[5], d[0]), lerp1_fused(v[6], v[7], d[0]), d[1]);
        return lerp1_fused(a, b, d[2]);
    }
}
template <typename T, int ND> __device__ __forceinline__ typename T::C interp_t(const typename T::C *v, const typename T::C *d) {
    if constexpr (sizeof(typename T::S) == 2) return interp_nd_fused<ND, typename T::C>(v, d);
    else return interp_nd<ND, typename T::C>(v, d);
}

// compute_weight_gradients (shifts_kernels.h:132-154), including the reference's 2-D "dx" wiring
// (interpolation.h:22-25: a difference along W blended with dW is used as the H-shift gradient).
template <int ND, typename CT> __device__ __forceinline__ void weight_grads_nd(const CT *v, const CT *d, CT *g) {
    if constexpr (ND == 1) {
        g[0] = v[1] - v[0];
    } else if constexpr (ND == 2) {
        g[0] = lerp1(v[2] - v[0], v[3] - v[1], d[1]);
        g[1] = lerp1(v[2], v[3], d[0]) - lerp1(v[0], v[1], d[0]);
    } else {
        g[0] = lerp1(lerp1(v[2] - v[0], v[3] - v[1], d[1]), lerp1(v[6] - v[4], v[7] - v[5], d[1]), d[2]);
        g[1] = lerp1(lerp1(v[2], v[3], d[0]) - lerp1(v[0], v[1], d[0]),
                     lerp1(v[6], v[7], d[0]) - lerp1(v[4], v[5], d[0]), d[2]);
        g[2] = lerp1(lerp1(v[4], v[5], d[0]), lerp1(v[6], v[7], d[0]), d[1]) -
               lerp1(lerp1(v[0], v[1], d[0]), lerp1(v[2], v[3], d[0]), d[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient in multilinear form.  Every partial of compute_weight_gradients is a multilinear blend,
// with per-channel (uniform) coefficients, of corner DIFFERENCES:
//   2-D: A = v2-v0, B = v3-v1:          wg0 = lerp(A, B, dW),  wg1 = lerp(A, B, dH)
//   3-D: A0 = v2-v0, B0 = v3-v1, A1 = v6-v4, B1 = v7-v5, Dq = v(q+4)-v(q):
//        wg0 = lerp(lerp(A0,B0,dW), lerp(A1,B1,dW), dD), wg1 = same with dH, wg2 = bilerp(D0..D3; dH, dW)
//   1-D: wg0 = v1-v0
// so sum_e g_e * wg_s(e) = blend_s(sum_e g_e * diff_k(e)): the streaming loop accumulates NDIFF sums of
// g * difference (differences first: no cancellation between large sums) and the blends are applied once per
// thread.  Same value as the per-element form up to fp32 rounding of each term (the tests' bar for grad_w is
// 1e-5 relative to an fp64 evaluation; on the dyadic fixture both forms are exact).
// ---------------------------------------------------------------------------------------------
template <int ND> struct WDiff { static constexpr int N = ND == 1 ? 1 : (ND == 2 ? 2 : 8); };

template <int ND, typename CT> __device__ __forceinline__ void corner_diffs(const CT *v, CT *d) {
    if constexpr (ND == 1) {
        d[0] = v[1] - v[0];
    } else if constexpr (ND == 2) {
        d[0] = v[2] - v[0];
        d[1] = v[3] - v[1];
    } else {
        d[0] = v[2] - v[0];
        d[1] = v[3] - v[1];
        d[2] = v[6] - v[4];
        d[3] = v[7] - v[5];
        d[4] = v[4] - v[0];
        d[5] = v[5] - v[1];
        d[6] = v[6] - v[2];
        d[7] = v[7] - v[3];
    }
}

// blends of the accumulated sums (fp64) -> the nD weight-gradient partials of this thread
template <int ND> __device__ __forceinline__ void blend_diffs(const double *s, const double *d /*dH,dW,dD*/, double *g) {
    auto lerp = [](double a, double b, double x) { return a * (1.0 - x) + b * x; };
    if constexpr (ND == 1) {
        g[0] = s[0];
    } else if constexpr (ND == 2) {
        g[0] = lerp(s[0], s[1], d[1]);
        g[1] = lerp(s[0], s[1], d[0]);
    } else {
        g[0] = lerp(lerp(s[0], s[1], d[1]), lerp(s[2], s[3], d[1]), d[2]);
        g[1] = lerp(lerp(s[0], s[1], d[0]), lerp(s[2], s[3], d[0]), d[2]);
        g[2] = lerp(lerp(s[4], s[5], d[0]), lerp(s[6], s[7], d[0]), d[1]);
    }
}

// v / cnt with IEEE rounding.  A power-of-two count (every window of an even pool size) divides exactly by
// multiplying with 2^-k: same bits as the division, none of its ~10 instructions.
template <typename CT> __device__ __forceinline__ CT div_count(CT v, int cnt) {
    if ((cnt & (cnt - 1)) != 0) return v / static_cast<CT>(cnt);
    const int k = __builtin_ctz(static_cast<unsigned>(cnt));
    CT scale;  // 2^-k from its bit pattern
    if constexpr (sizeof(CT) == 4) {
        const uint32_t bits = static_cast<uint32_t>(127 - k) << 23;
        __builtin_memcpy(&scale, &bits, 4);
    } else {
        const uint64_t bits = static_cast<uint64_t>(1023 - k) << 52;
        __builtin_memcpy(&scale, &bits, 8);
    }
    return v * scale;
}

// ---------------------------------------------------------------------------------------------
// Workgroup-wide fp64 sum (deterministic: fixed shuffle tree, fixed wave order).
// Returns the total in thread 0.  `scratch` needs kThreads/64 doubles of LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];
    }
    return t;
}

// Final stage of the weight-gradient reduction: grad_w[c, s] = sum over partial groups.
// partials layout: [group][C][3] doubles.  One copy of the kernels in the library (shiftnd_strided.hip defines them and this
// launcher); T::kDtype selects the output type.
void launch_reduce_weight_grads(int dtype, const double *partials, int groups, int C, int nd, void *grad_w, hipStream_t st);
template <typename T>
inline void reduce_weight_grads_of(const double *partials, int groups, int C, int nd, void *grad_w, hipStream_t st) {
    launch_reduce_weight_grads(T::kDtype, partials, groups, C, nd, grad_w, st);
}

// ---------------------------------------------------------------------------------------------
// buffer_store_dwordx4 with the row / plane offset in an SGPR (soffset).  gfx950 reads the 128 bits of store data over more than
// one cycle: a VALU instruction that overwrites one of the data registers in the very next issue slot corrupts what is stored.
// LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) inserts the wait state for a MUBUF store of more than 64 bits
// only when soffset is NOT a register; with a register soffset nothing separated
//     buffer_store_dwordx4 v[2:5], v73, s[24:27], s56 offen nt
//     v_cndmask_b32_e64 v2, 0, -1, s[16:17]          <- loop bookkeeping the scheduler placed in the store's shadow
// in walk_backward16<.., ZEROS = false> (hipcc 7.2, -O3): non-deterministic zeros in grad_x whenever the memory pipeline was busy
// enough to delay the data read -- several workgroups per CU -- and never on small shapes.  Round 4 took it for an LDS race and
// "fixed" it with waits and compiler barriers around __syncthreads(), which moved the v_cndmask 14 instructions away by accident.
// Round 5: tools/isa_barriers.py + an instruction diff of the two builds found the pair; a build with bare barriers and
// `s_nop 1` behind the store is clean on the shapes the bare build fails 20 times out of 20 (tools/race_probe.py;
// profiles/r05_walk_race.txt).  The asm's input operand keeps the data registers live up to the wait states.
// SHIFTND_DIAG_NO_STORE_NOP reproduces the failure.
// ---------------------------------------------------------------------------------------------
typedef uint32_t shiftnd_u4 __attribute__((ext_vector_type(4)));
template <int AUX>   // cache-policy bits of the instruction (2 = nontemporal)
__device__ __forceinline__ void buffer_store_b128_soffset(shiftnd_u4 data, __amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset, soffset, AUX);
#ifndef SHIFTND_DIAG_NO_STORE_NOP
    asm volatile("s_nop 1" ::"v"(data));
#endif
}

// ---------------------------------------------------------------------------------------------
// Raw element / chunk I/O shared by the kernel families.
// ---------------------------------------------------------------------------------------------
template <int ESIZE> struct raw_t;
template <> struct raw_t<1> { using type = uint8_t; };
template <> struct raw_t<2> { using type = uint16_t; };
template <> struct raw_t<4> { using type = uint32_t; };
template <> struct raw_t<8> { using type = uint64_t; };

template <int V> struct vec_of;
template <> struct vec_of<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct vec_of<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct vec_of<4> { typedef uint32_t type; };
template <> struct vec_of<2> { typedef uint16_t type; };
template <> struct vec_of<1> { typedef uint8_t type; };

template <typename R, int E> struct Chunk { R e[E]; };  // one V = sizeof(R) * E byte piece of a row

// element-aligned V-byte load (one global_load_dwordx4 for V = 16: gfx950 global loads take any alignment).
// NT = nontemporal: only for data that no other workgroup re-reads (the gather forward); the backward kernels
// rely on L2 for the rows that neighbouring rows share (measured: nontemporal loads there raise HBM reads from
// 7.6 to 10 GB per C2 launch).
template <typename R, int E, bool NT = false> __device__ __forceinline__ Chunk<R, E> load_chunk(const R *src) {
    constexpr int V = sizeof(R) * E;
    typedef typename vec_of<V>::type vec_t;
    typedef vec_t unaligned_t __attribute__((aligned(sizeof(R) < 4 ? sizeof(R) : 4)));
    const vec_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const unaligned_t *>(src))
                       : *reinterpret_cast<const unaligned_t *>(src);
    Chunk<R, E> c;
    __builtin_memcpy(c.e, &v, V);
    return c;
}
// element-aligned V-byte nontemporal store (one global_store_dwordx4 for V = 16 at a 4-byte boundary)
// (the caller guarantees a 4-byte boundary also for 2-byte elements: rows of an even number of them)
template <typename R, int E> __device__ __forceinline__ void store_chunk_unaligned(R *dst, const Chunk<R, E> &c) {
    constexpr int V = sizeof(R) * E;
    typedef typename vec_of<V>::type vec_t;
    typedef vec_t unaligned_t __attribute__((aligned(4)));
    vec_t v;
    __builtin_memcpy(&v, c.e, V);
    __builtin_nontemporal_store(v, reinterpret_cast<unaligned_t *>(dst));
}
// aligned V-byte nontemporal store (every output byte is written once)
template <typename R, int E> __device__ __forceinline__ void store_chunk(R *dst, const Chunk<R, E> &c) {
    constexpr int V = sizeof(R) * E;
    typedef typename vec_of<V>::type vec_t;
    vec_t v;
    __builtin_memcpy(&v, c.e, V);
    __builtin_nontemporal_store(v, reinterpret_cast<vec_t *>(dst));
}

// float weight of any supported dtype, widened to the compute type
template <typename CT> __device__ __forceinline__ CT load_weight(const void *w, int wkind, int64_t i) {
    switch (wkind) {
    case SHIFTND_F64: return static_cast<CT>(static_cast<const double *>(w)[i]);
    case SHIFTND_F16: return static_cast<CT>(static_cast<const _Float16 *>(w)[i]);
    case SHIFTND_BF16: return static_cast<CT>(static_cast<const __bf16 *>(w)[i]);
    default: return static_cast<CT>(static_cast<const float *>(w)[i]);
    }
}

// The (up to) three weights of a channel, every load issued before the first conversion: load_weight() per dim pays one
// memory round trip per dim in the workgroup prologue.  Dims without a weight column read column 0 (never used).
template <typename CT> __device__ __forceinline__ void load_weights3(const void *w, int wkind, int64_t base, const int wcol[3], CT out[3]) {
    const int64_t i0 = base + (wcol[0] > 0 ? wcol[0] : 0), i1 = base + (wcol[1] > 0 ? wcol[1] : 0), i2 = base + (wcol[2] > 0 ? wcol[2] : 0);
#define SHIFTND_LOAD3(TYPE) \
    { \
        const TYPE *q = static_cast<const TYPE *>(w); \
        const TYPE r0 = q[i0], r1 = q[i1], r2 = q[i2]; \
        out[0] = static_cast<CT>(r0); \
        out[1] = static_cast<CT>(r1); \
        out[2] = static_cast<CT>(r2); \
    } \
    break;
    switch (wkind) {
    case SHIFTND_F64: SHIFTND_LOAD3(double)
    case SHIFTND_F16: SHIFTND_LOAD3(_Float16)
    case SHIFTND_BF16: SHIFTND_LOAD3(__bf16)
    default: SHIFTND_LOAD3(float)
    }
#undef SHIFTND_LOAD3
}

// integer shift of the gather-only kernels: round-half-even of a float weight (SSL), or a quantized weight's
// int_repr minus its zero point (kernels/shifts_kernels.h:553-555)
__device__ __forceinline__ int64_t gather_shift(const void *w, int wkind, int64_t wzp, int64_t i) {
    switch (wkind) {
    case SHIFTND_F32: return static_cast<int64_t>(rintf(static_cast<const float *>(w)[i]));
    case SHIFTND_F64: return static_cast<int64_t>(rint(static_cast<const double *>(w)[i]));
    case SHIFTND_F16: return static_cast<int64_t>(rintf(static_cast<float>(static_cast<const _Float16 *>(w)[i])));
    case SHIFTND_BF16: return static_cast<int64_t>(rintf(static_cast<float>(static_cast<const __bf16 *>(w)[i])));
    case SHIFTND_I8: return static_cast<int64_t>(static_cast<const int8_t *>(w)[i]) - wzp;
    case SHIFTND_U8: return static_cast<int64_t>(static_cast<const uint8_t *>(w)[i]) - wzp;
    default: return static_cast<int64_t>(static_cast<const int32_t *>(w)[i]) - wzp;
    }
}

// gather_shift for the (up to) three dims of a channel with the loads issued together (see load_weights3)
__device__ __forceinline__ void gather_shifts3(const void *w, int wkind, int64_t wzp, int64_t base, const int wcol[3], int64_t out[3]) {
    const int64_t i0 = base + (wcol[0] > 0 ? wcol[0] : 0), i1 = base + (wcol[1] > 0 ? wcol[1] : 0), i2 = base + (wcol[2] > 0 ? wcol[2] : 0);
#define SHIFTND_GATHER3(TYPE, EXPR) \
    { \
        const TYPE *q = static_cast<const TYPE *>(w); \
        const TYPE r0 = q[i0], r1 = q[i1], r2 = q[i2]; \
        { const TYPE r = r0; out[0] = (EXPR); } \
        { const TYPE r = r1; out[1] = (EXPR); } \
        { const TYPE r = r2; out[2] = (EXPR); } \
    } \
    break;
    switch (wkind) {
    case SHIFTND_F32: SHIFTND_GATHER3(float, static_cast<int64_t>(rintf(r)))
    case SHIFTND_F64: SHIFTND_GATHER3(double, static_cast<int64_t>(rint(r)))
    case SHIFTND_F16: SHIFTND_GATHER3(_Float16, static_cast<int64_t>(rintf(static_cast<float>(r))))
    case SHIFTND_BF16: SHIFTND_GATHER3(__bf16, static_cast<int64_t>(rintf(static_cast<float>(r))))
    case SHIFTND_I8: SHIFTND_GATHER3(int8_t, static_cast<int64_t>(r) - wzp)
    case SHIFTND_U8: SHIFTND_GATHER3(uint8_t, static_cast<int64_t>(r) - wzp)
    default: SHIFTND_GATHER3(int32_t, static_cast<int64_t>(r) - wzp)
    }
#undef SHIFTND_GATHER3
}

// ---------------------------------------------------------------------------------------------
// Host-side problem description handed to the kernel families.
// ---------------------------------------------------------------------------------------------
struct Geometry {
    int nd;
    int pad;
    int active;
    int64_t N, C;
    int64_t S[3];    // input spatial sizes, normalised to 3 dims (leading dims = 1): [d0][d1][inner]
    int64_t O[3];    // output (forward) / grad_out (backward) spatial sizes
    int64_t L[3];    // left borders
    int64_t xs[5];   // element strides of x, in normalised order N, C, d0, d1, inner
    int64_t os[5];   // element strides of out (forward) / grad_out (backward)
    int64_t gs[5];   // backward: element strides of grad_x
    int wcol[3];     // weight column of each normalised dim, or -1
    int64_t K[3];    // fused average pool: window = stride per normalised dim (1 = none); 0 when the call has no pool
    int64_t P[3];    // pooled sizes ceil(O / K)
};

}  // namespace shiftnd
