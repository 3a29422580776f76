// shiftnd_transpose.hip -- batched matrix transpose, the layout change between channels-last and contiguous
// tensors: dst[n][c][r] = src[n][r][c] for dense src[N][rows][cols], dst[N][cols][rows].
//
// Why it exists: the reference's float ops return an NCHW-contiguous tensor even for a channels-last input
// (cpu/shifts_cpu.cpp:221) and its CUDA backend walks a channels-last input through strides (uncoalesced).  With
// mixed layouts one side of a gather kernel is uncoalesced whichever way it iterates; changing the
// layout first with a tile transpose at copy bandwidth and running the contiguous kernels is several times faster.
//
// One workgroup moves a 64 x 64 element tile through LDS: 16-byte global loads along `cols`, 16-byte global stores
// along `rows`; the LDS row pitch is odd in dwords so the column-wise reads are bank-conflict free.  Ragged tiles
// and shapes whose rows are not whole 16-byte pieces take the element-wise path.
#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

constexpr int kTile = 64;

template <int ESIZE, bool VEC>
__global__ __launch_bounds__(kThreads) void transpose_tiles(const void *__restrict__ src_, void *__restrict__ dst_, int rows,
                                                            int cols, int tiles_r, int tiles_c) {
    using R = typename raw_t<ESIZE>::type;
    constexpr int E = 16 / ESIZE;                       // elements per 16-byte piece
    // bytes; the extra dword (two for 8-byte elements, whose LDS accesses must stay 8-byte aligned) staggers the banks
    constexpr int PITCH = kTile * ESIZE + (ESIZE == 8 ? 8 : 4);
    __shared__ __attribute__((aligned(16))) char tile[kTile * PITCH + 16];
    // consecutive workgroups walk the SHORTER tile dimension first (the channel dimension of either direction), so
    // the long contiguous side -- whole channels-last pixel rows -- is read or written as one contiguous region by
    // neighbouring workgroups (NCHW -> channels-last N16 C256 224x224 fp32: 1.37 -> see DESIGN.md)
    int b = blockIdx.x, tr, tc;
    if (tiles_r < tiles_c) {
        tr = b % tiles_r;
        b /= tiles_r;
        tc = b % tiles_c;
        b /= tiles_c;
    } else {
        tc = b % tiles_c;
        b /= tiles_c;
        tr = b % tiles_r;
        b /= tiles_r;
    }
    const int n = b;
    const int r0 = tr * kTile, c0 = tc * kTile;
    const R *src = static_cast<const R *>(src_) + static_cast<int64_t>(n) * rows * cols;
    R *dst = static_cast<R *>(dst_) + static_cast<int64_t>(n) * rows * cols;
    const bool full = r0 + kTile <= rows && c0 + kTile <= cols;
    if (VEC && full) {
        constexpr int VPR = kTile / E;                  // 16-byte pieces per tile row
#pragma unroll
        for (int k = 0; k < kTile * VPR / kThreads; ++k) {
            const int v = k * kThreads + static_cast<int>(threadIdx.x);
            const int r = v / VPR, cv = v - r * VPR;
            const Chunk<R, E> ch = load_chunk<R, E, true>(src + static_cast<int64_t>(r0 + r) * cols + c0 + cv * E);
            // the pitch is not a multiple of 16: element-size stores
#pragma unroll
            for (int e = 0; e < E; ++e) *reinterpret_cast<R *>(tile + r * PITCH + (cv * E + e) * ESIZE) = ch.e[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kTile * VPR / kThreads; ++k) {
            const int v = k * kThreads + static_cast<int>(threadIdx.x);
            const int c = v / VPR, rv = v - c * VPR;   // output row c of the tile, piece rv along the source rows
            Chunk<R, E> ch;
#pragma unroll
            for (int e = 0; e < E; ++e) ch.e[e] = *reinterpret_cast<const R *>(tile + (rv * E + e) * PITCH + c * ESIZE);
            store_chunk<R, E>(dst + static_cast<int64_t>(c0 + c) * rows + r0 + rv * E, ch);
        }
    } else {
        for (int v = threadIdx.x; v < kTile * kTile; v += kThreads) {
            const int r = v / kTile, c = v - r * kTile;
            if (r0 + r < rows && c0 + c < cols)
                *reinterpret_cast<R *>(tile + r * PITCH + c * ESIZE) = src[static_cast<int64_t>(r0 + r) * cols + c0 + c];
        }
        __syncthreads();
        for (int v = threadIdx.x; v < kTile * kTile; v += kThreads) {
            const int c = v / kTile, r = v - c * kTile;
            if (r0 + r < rows && c0 + c < cols)
                dst[static_cast<int64_t>(c0 + c) * rows + r0 + r] = *reinterpret_cast<const R *>(tile + r * PITCH + c * ESIZE);
        }
    }
}

template <int ESIZE>
int launch_transpose(const void *src, void *dst, int64_t N, int64_t rows, int64_t cols, hipStream_t st) {
    const int64_t tr = (rows + kTile - 1) / kTile, tc = (cols + kTile - 1) / kTile;
    const int64_t blocks = N * tr * tc;
    if (blocks <= 0) return SHIFTND_OK;
    if (blocks >= (1LL << 31) || rows >= (1LL << 31) || cols >= (1LL << 31)) return SHIFTND_ERR_TOO_LARGE;
    const bool vec = (rows * ESIZE) % 16 == 0 && (cols * ESIZE) % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(dst) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL((transpose_tiles<ESIZE, true>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, src, dst,
                           static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(tr), static_cast<int>(tc));
    else
        hipLaunchKernelGGL((transpose_tiles<ESIZE, false>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, src, dst,
                           static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(tr), static_cast<int>(tc));
    return SHIFTND_OK;
}

}  // namespace

int transpose_planes(const void *src, void *dst, int64_t N, int64_t rows, int64_t cols, int esize, hipStream_t st) {
    note_kernel("transpose_tiles");
    switch (esize) {
    case 1: return launch_transpose<1>(src, dst, N, rows, cols, st);
    case 2: return launch_transpose<2>(src, dst, N, rows, cols, st);
    case 4: return launch_transpose<4>(src, dst, N, rows, cols, st);
    case 8: return launch_transpose<8>(src, dst, N, rows, cols, st);
    default: return SHIFTND_ERR_UNSUPPORTED_DTYPE;
    }
}

}  // namespace shiftnd
