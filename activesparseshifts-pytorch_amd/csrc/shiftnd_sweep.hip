// shiftnd_sweep.hip -- "sweep" kernels: the HBM-rate path of the shiftnd op on gfx950 (MI355X).
//
// Measured on MI355X (tools/hbm_bench.hip): a wave that performs ONE round of loads followed by ONE
// 16-byte store and then retires streams at 6.4-6.6 TB/s, while workgroups that loop over a private
// region (one plane each) saturate at 5.2-5.5 TB/s however the loop is pipelined.  The sweep kernels
// therefore give every thread exactly one 16-byte output chunk:
//   * the grid is the list of output chunks in memory order, 256 per workgroup, no loops, no LDS maps,
//     no barriers in the forward kernels;
//   * blockIdx is remapped so that the workgroups of one XCD (blockIdx % 8) cover a contiguous eighth
//     of the tensor: rows re-read by neighbouring workgroups (interpolation corners, the shifted
//     grad_out row) hit that XCD's L2 instead of travelling twice from HBM;
//   * loads and stores are nontemporal (every byte is touched once);
//   * the padding map is evaluated arithmetically per element: the channel's shift is reduced to a
//     canonical representative once per thread (canon_shift), after which any index needs at most
//     two conditional folds (fold_index) -- exact for all five modes and any shift magnitude;
//   * a chunk whose E (+1) source columns are consecutive takes one element-aligned 16-byte load,
//     otherwise (chunks touching the padded region) E masked element loads.
//
// Reference behaviour restated (paths under torchshifts/csrc/ops/):
//   forward   kernels/shifts_kernels.h:156-220, cuda/shifts_cuda.cu:202-266
//   backward  kernels/shifts_kernels.h:222-327, cuda/shifts_cuda.cu:270-345
//   quantized kernels/shifts_kernels.h:532-571, quantized/shifts_quantized.cpp:107-130
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

struct SweepParams {
    const void *x;     // forward: input; backward: saved input
    const void *go;    // backward: incoming gradient
    void *out;         // forward: output; backward: grad_x
    const void *w;
    double *partials;  // backward: [blocks_per_plane * N][C][3]
    int64_t wzp;
    uint64_t fill;
    int64_t x_plane, o_plane;  // elements per (n, c) plane
    uint32_t total;            // forward: chunks in the whole tensor; backward: chunks per plane
    uint32_t blocks;           // logical workgroups
    uint32_t blocks_per_xcd;   // ceil(blocks / 8)
    uint32_t bpp;              // backward: workgroups per plane
    int wkind, C, nd, pad;
    int S[3], O[3], L[3], wcol[3];
    uint32_t cpr, cpp;         // chunks per row / per plane of the iteration space
    FastDiv d_cpp, d_cpr, d_dim1, d_C, d_bpp;
    FastDiv d_per[3];          // divide by the padding period of each dim of x
    FastDiv d_gper[3];         // backward: same for the grad_out dims
};

// workgroups that share an XCD (blockIdx % 8, round-robin dispatch) get consecutive logical ids
__device__ __forceinline__ uint32_t xcd_remap(uint32_t blocks_per_xcd) {
    return (blockIdx.x & 7u) * blocks_per_xcd + (blockIdx.x >> 3);
}

template <int ESIZE> struct raw_t;
template <> struct raw_t<1> { using type = uint8_t; };
template <> struct raw_t<2> { using type = uint16_t; };
template <> struct raw_t<4> { using type = uint32_t; };
template <> struct raw_t<8> { using type = uint64_t; };

// V-byte vectors with element alignment for loads (gfx950 global loads take any alignment)
template <int V> struct vec_of;
template <> struct vec_of<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct vec_of<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct vec_of<4> { typedef uint32_t type; };
template <> struct vec_of<2> { typedef uint16_t type; };
template <> struct vec_of<1> { typedef uint8_t type; };

template <typename R, int E> struct Chunk { R e[E]; };

template <typename R, int E> __device__ __forceinline__ Chunk<R, E> load_chunk_nt(const R *src) {
    constexpr int V = sizeof(R) * E;
    typedef typename vec_of<V>::type vec_t;
    typedef vec_t unaligned_t __attribute__((aligned(sizeof(R) < 4 ? sizeof(R) : 4)));
    const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const unaligned_t *>(src));
    Chunk<R, E> c;
    __builtin_memcpy(c.e, &v, V);
    return c;
}
template <typename R, int E> __device__ __forceinline__ void store_chunk_nt(R *dst, const Chunk<R, E> &c) {
    constexpr int V = sizeof(R) * E;
    typedef typename vec_of<V>::type vec_t;
    vec_t v;
    __builtin_memcpy(&v, c.e, V);
    __builtin_nontemporal_store(v, reinterpret_cast<vec_t *>(dst));
}
template <typename R> __device__ __forceinline__ R load_elem_nt(const R *src) { return __builtin_nontemporal_load(src); }

__device__ __forceinline__ int64_t gather_shift(const void *w, int wkind, int64_t wzp, int i) {
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

// source index of coordinate p of a dim: -1 = fill
__device__ __forceinline__ int map1(int p, int cs, int len, int pad) { return len == 1 ? 0 : fold_index(p - cs, len, pad); }

// =====================================================================================================
// Gather forward (SSL forward of every float dtype, quantized forward)
// =====================================================================================================
template <int ESIZE, int V>
__global__ __launch_bounds__(kThreads) void sweep_gather_forward(const SweepParams p) {
    using R = typename raw_t<ESIZE>::type;
    constexpr int E = V / ESIZE;
    // ---- workgroup-uniform part (scalar registers): which plane, which channel, its canonical shifts ----
    const uint32_t bid = xcd_remap(p.blocks_per_xcd);
    if (bid >= p.blocks) return;
    const uint32_t plane = fdiv(bid, p.d_bpp);
    const uint32_t blk = bid - plane * p.bpp;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        cs[d] = p.wcol[d] >= 0 ? canon_shift(gather_shift(p.w, p.wkind, p.wzp, c * p.nd + p.wcol[d]), p.S[d], p.pad, p.d_per[d]) : 0;
    // ---- per-thread part: chunk -> (row, column) ---------------------------------------------------------
    const uint32_t q = blk * kThreads + threadIdx.x;
    if (q >= p.cpp) return;
    const uint32_t r = fdiv(q, p.d_cpr);
    const int jo = static_cast<int>(q - r * p.cpr) * E;
    const uint32_t a = fdiv(r, p.d_dim1);
    const uint32_t b = r - a * static_cast<uint32_t>(p.O[1]);

    const R fill = static_cast<R>(p.fill);
    R *dst = static_cast<R *>(p.out) + static_cast<int64_t>(plane) * p.o_plane + static_cast<int64_t>(r) * p.O[2] + jo;
    Chunk<R, E> v;
    const int ra = map1(static_cast<int>(a) + p.L[0], cs[0], p.S[0], p.pad);
    const int rb = map1(static_cast<int>(b) + p.L[1], cs[1], p.S[1], p.pad);
    if (ra < 0 || rb < 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) v.e[e] = fill;
    } else {
        const R *row = static_cast<const R *>(p.x) + static_cast<int64_t>(plane) * p.x_plane +
                       static_cast<int64_t>(ra * p.S[1] + rb) * p.S[2];
        int mm[E];
        bool contig = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mm[e] = map1(jo + p.L[2] + e, cs[2], p.S[2], p.pad);
            contig = contig && (mm[e] == mm[0] + e);
        }
        if (contig && mm[0] >= 0) {
            v = load_chunk_nt<R, E>(row + mm[0]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) v.e[e] = mm[e] >= 0 ? load_elem_nt<R>(row + mm[e]) : fill;
        }
    }
    store_chunk_nt<R, E>(dst, v);
}

// =====================================================================================================
// Host side
// =====================================================================================================
bool contiguous(const int64_t st[5], int64_t N, int64_t C, const int64_t sz[3]) {
    int64_t expect = 1;
    const int64_t sizes[5] = {N, C, sz[0], sz[1], sz[2]};
    for (int d = 4; d >= 0; --d) {
        if (sizes[d] != 1 && st[d] != expect) return false;
        expect *= sizes[d];
    }
    return true;
}

int gather_vector_bytes(const Geometry &g, int esize, const void *out) {
    const int cand[3] = {16, 8, 4};
    for (int V : cand) {
        if (V < esize) continue;
        if ((g.O[2] * esize) % V != 0) continue;
        if (reinterpret_cast<uintptr_t>(out) % V != 0) continue;
        return V;
    }
    return esize;
}

void fill_common(SweepParams &p, const Geometry &g) {
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.O[d] = static_cast<int>(g.O[d]);
        p.L[d] = static_cast<int>(g.L[d]);
        p.wcol[d] = g.wcol[d];
    }
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.o_plane = g.O[0] * g.O[1] * g.O[2];
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    for (int d = 0; d < 3; ++d) {
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], p.pad)));
        p.d_gper[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.O[d], p.pad)));
    }
}

void set_grid(SweepParams &p, uint64_t blocks) {
    p.blocks = static_cast<uint32_t>(blocks);
    p.blocks_per_xcd = static_cast<uint32_t>((blocks + 7) / 8);
}

template <int ESIZE, int V> void launch_gather(const SweepParams &p, hipStream_t st) {
    hipLaunchKernelGGL((sweep_gather_forward<ESIZE, V>), dim3(p.blocks_per_xcd * 8), dim3(kThreads), 0, st, p);
}

}  // namespace

bool sweep_forward_eligible(const Geometry &g, int dtype, const void *x, const void *out) {
    (void)x;
    (void)out;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], oe = g.O[0] * g.O[1] * g.O[2];
    if (xe >= (1LL << 30) || oe >= (1LL << 30)) return false;  // 32-bit in-plane offsets
    if (g.N * g.C >= (1LL << 31)) return false;
    if (!contiguous(g.xs, g.N, g.C, g.S) || !contiguous(g.os, g.N, g.C, g.O)) return false;
    const bool interpolating = g.active && dtype <= SHIFTND_BF16;
    if (interpolating) return false;  // active forward: plane kernels (for now)
    const int es = dtype_size(dtype);
    const int V = gather_vector_bytes(g, es, out);
    const int64_t cpp = oe * es / V;
    const int64_t blocks = g.N * g.C * ((cpp + kThreads - 1) / kThreads);
    return blocks < (1LL << 31) - 16;  // 32-bit workgroup ids (also the grid limit)
}

int sweep_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, uint64_t fill_bits,
                  void *out, hipStream_t st) {
    const int es = dtype_size(dtype);
    const int V = gather_vector_bytes(g, es, out);
    SweepParams p{};
    fill_common(p, g);
    p.x = x;
    p.out = out;
    p.w = w;
    p.wkind = wkind;
    p.wzp = wzp;
    p.fill = fill_bits;
    p.cpr = static_cast<uint32_t>(g.O[2] * es / V);
    p.cpp = static_cast<uint32_t>(g.O[0] * g.O[1]) * p.cpr;
    p.bpp = (p.cpp + kThreads - 1) / kThreads;
    p.d_bpp = make_fastdiv(p.bpp);
    p.d_cpr = make_fastdiv(p.cpr);
    p.d_dim1 = make_fastdiv(static_cast<uint32_t>(g.O[1]));
    set_grid(p, static_cast<uint64_t>(g.N * g.C) * p.bpp);
#define SHIFTND_GATHER_CASE(ES, VV) \
    if (es == ES && V == VV) { launch_gather<ES, VV>(p, st); return SHIFTND_OK; }
    SHIFTND_GATHER_CASE(1, 16) SHIFTND_GATHER_CASE(1, 8) SHIFTND_GATHER_CASE(1, 4) SHIFTND_GATHER_CASE(1, 1)
    SHIFTND_GATHER_CASE(2, 16) SHIFTND_GATHER_CASE(2, 8) SHIFTND_GATHER_CASE(2, 4) SHIFTND_GATHER_CASE(2, 2)
    SHIFTND_GATHER_CASE(4, 16) SHIFTND_GATHER_CASE(4, 8) SHIFTND_GATHER_CASE(4, 4)
    SHIFTND_GATHER_CASE(8, 16) SHIFTND_GATHER_CASE(8, 8)
#undef SHIFTND_GATHER_CASE
    return SHIFTND_ERR_UNSUPPORTED_DTYPE;
}

// host mirror of the per-element map, for tests: source index of coordinate p (or -1)
int sweep_debug_map(int64_t p, int64_t shift, int64_t len, int pad) {
    if (len == 1) return 0;
    const FastDiv dper = make_fastdiv(static_cast<uint32_t>(map_period(static_cast<int>(len), pad)));
    return fold_index(static_cast<int>(p) - canon_shift(shift, static_cast<int>(len), pad, dper), static_cast<int>(len), pad);
}

}  // namespace shiftnd
