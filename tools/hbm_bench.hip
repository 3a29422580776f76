// hbm_bench.hip -- what can this box's HBM actually sustain?  (diagnostic, not part of the product)
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_bench.hip -o tools/hbm_bench && tools/hbm_bench
// Streams 3.29 GB buffers (the C2 tensor size) with float4 accesses: copy (1R:1W), 2R:1W (the backward
// kernel's mix), read-only and write-only, with plain / nontemporal accesses and several grid shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int U>  // MODE bit0: nt store, bit1: nt load
__global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ a, f4 *__restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (MODE & 2) ? __builtin_nontemporal_load(&a[i + u * stride]) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE & 1) __builtin_nontemporal_store(v[u], &o[i + u * stride]);
            else o[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) o[i] = a[i];
}

// contiguous chunk per block (each block streams its own region, like one plane per workgroup)
template <int MODE>
__global__ __launch_bounds__(256) void copy_blocked(const f4 *__restrict__ a, f4 *__restrict__ o, size_t per_block) {
    const f4 *ap = a + (size_t)blockIdx.x * per_block;
    f4 *op = o + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 256 * 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < per_block) v[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[i + u * 256]) : ap[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < per_block) {
            if (MODE & 1) __builtin_nontemporal_store(v[u], &op[i + u * 256]);
            else op[i + u * 256] = v[u];
        }
    }
}

// one-shot blocked 2R:1W: block b owns elements [b*per_block, (b+1)*per_block) of all three arrays
template <int MODE>
__global__ __launch_bounds__(256) void add2_blocked(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o,
                                                    size_t per_block, int remap) {
    size_t blk = blockIdx.x;
    if (remap) {  // XCD-contiguous: blocks that share an XCD (same blockIdx % 8) get consecutive chunks
        const size_t per_xcd = gridDim.x / 8;
        blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    }
    const f4 *ap = a + blk * per_block;
    const f4 *bp = b + blk * per_block;
    f4 *op = o + blk * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 256 * 2) {
        f4 x[2], y[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) if (i + u * 256 < per_block) {
            x[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[i + u * 256]) : ap[i + u * 256];
            y[u] = (MODE & 2) ? __builtin_nontemporal_load(&bp[i + u * 256]) : bp[i + u * 256];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) if (i + u * 256 < per_block) {
            f4 r = x[u] + y[u];
            if (MODE & 1) __builtin_nontemporal_store(r, &op[i + u * 256]);
            else op[i + u * 256] = r;
        }
    }
}

// software-pipelined persistent loops: the next iteration's loads are issued BEFORE this iteration's
// stores, so waiting for them (in-order vmcnt) does not wait for the stores to be acknowledged
template <int MODE, int U>
__global__ __launch_bounds__(256) void copy_pipe(const f4 *__restrict__ a, f4 *__restrict__ o, size_t per_block) {
    const f4 *ap = a + (size_t)blockIdx.x * per_block;
    f4 *op = o + (size_t)blockIdx.x * per_block;
    f4 cur[U], nxt[U];
    size_t i = threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < per_block) cur[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[i + u * 256]) : ap[i + u * 256];
    for (; i < per_block; i += 256 * U) {
        const size_t j = i + 256 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) if (j + u * 256 < per_block) nxt[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[j + u * 256]) : ap[j + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < per_block) {
            if (MODE & 1) __builtin_nontemporal_store(cur[u], &op[i + u * 256]);
            else op[i + u * 256] = cur[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

template <int MODE, int U>
__global__ __launch_bounds__(256) void add2_pipe(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, size_t per_block) {
    const f4 *ap = a + (size_t)blockIdx.x * per_block;
    const f4 *bp = b + (size_t)blockIdx.x * per_block;
    f4 *op = o + (size_t)blockIdx.x * per_block;
    f4 cx[U], cy[U], nx[U], ny[U];
    size_t i = threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < per_block) {
        cx[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[i + u * 256]) : ap[i + u * 256];
        cy[u] = (MODE & 2) ? __builtin_nontemporal_load(&bp[i + u * 256]) : bp[i + u * 256];
    }
    for (; i < per_block; i += 256 * U) {
        const size_t j = i + 256 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) if (j + u * 256 < per_block) {
            nx[u] = (MODE & 2) ? __builtin_nontemporal_load(&ap[j + u * 256]) : ap[j + u * 256];
            ny[u] = (MODE & 2) ? __builtin_nontemporal_load(&bp[j + u * 256]) : bp[j + u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < per_block) {
            f4 r = cx[u] + cy[u];
            if (MODE & 1) __builtin_nontemporal_store(r, &op[i + u * 256]);
            else op[i + u * 256] = r;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; }
    }
}

// one-shot: block of T threads copies T*K float4 (K loads then K stores per thread, no loop)
template <int T, int K, int MODE>
__global__ __launch_bounds__(T) void copy_oneshot(const f4 *__restrict__ a, f4 *__restrict__ o, int remap) {
    size_t blk = blockIdx.x;
    if (remap) {
        const size_t per_xcd = gridDim.x / 8;
        blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    }
    const size_t base = blk * (size_t)(T * K) + threadIdx.x;
    f4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (MODE & 2) ? __builtin_nontemporal_load(&a[base + k * T]) : a[base + k * T];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (MODE & 1) __builtin_nontemporal_store(v[k], &o[base + k * T]);
        else o[base + k * T] = v[k];
    }
}

template <int T, int K, int MODE>
__global__ __launch_bounds__(T) void add2_oneshot(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, int remap) {
    size_t blk = blockIdx.x;
    if (remap) {
        const size_t per_xcd = gridDim.x / 8;
        blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    }
    const size_t base = blk * (size_t)(T * K) + threadIdx.x;
    f4 x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        x[k] = (MODE & 2) ? __builtin_nontemporal_load(&a[base + k * T]) : a[base + k * T];
        y[k] = (MODE & 2) ? __builtin_nontemporal_load(&b[base + k * T]) : b[base + k * T];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        f4 r = x[k] + y[k];
        if (MODE & 1) __builtin_nontemporal_store(r, &o[base + k * T]);
        else o[base + k * T] = r;
    }
}

// persistent grid-stride, software pipelined (single sweep front, chunk = 256 float4 per block per step)
template <int MODE>
__global__ __launch_bounds__(256) void copy_gs_pipe(const f4 *__restrict__ a, f4 *__restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
    if (i < n) cur = (MODE & 2) ? __builtin_nontemporal_load(&a[i]) : a[i];
    for (; i < n; i += stride) {
        if (i + stride < n) nxt = (MODE & 2) ? __builtin_nontemporal_load(&a[i + stride]) : a[i + stride];
        if (MODE & 1) __builtin_nontemporal_store(cur, &o[i]);
        else o[i] = cur;
        cur = nxt;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void add2_k(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        f4 x = (MODE & 2) ? __builtin_nontemporal_load(&a[i]) : a[i];
        f4 y = (MODE & 2) ? __builtin_nontemporal_load(&b[i]) : b[i];
        f4 r = x + y;
        if (MODE & 1) __builtin_nontemporal_store(r, &o[i]);
        else o[i] = r;
    }
}

__global__ __launch_bounds__(256) void read_k(const f4 *__restrict__ a, float *__restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) o[0] = 1.f;
}

template <int MODE>
__global__ __launch_bounds__(256) void write_k(f4 *__restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE & 1) __builtin_nontemporal_store(v, &o[i]);
        else o[i] = v;
    }
}

template <typename F> float time_ms(F f, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) f();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best) best = ms / iters;
    }
    return best;
}

int main() {
    const size_t bytes = 64ull * 256 * 224 * 224 * 4;
    const size_t n = bytes / 16;
    f4 *a, *b, *o;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&o, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    CHECK(hipMemset(o, 0, bytes));
    const double gb = bytes / 1e9;
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 64, 16384, (int)((n + 255) / 256)};
    for (int g : grids) {
        float t;
        t = time_ms([&] { hipLaunchKernelGGL((copy_k<0, 4>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        printf("copy plain      U4 grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((copy_k<1, 4>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        printf("copy nt-store   U4 grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((copy_k<3, 4>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        printf("copy nt-ld+st   U4 grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((copy_k<0, 1>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        printf("copy plain      U1 grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((add2_k<0>), dim3(g), dim3(256), 0, 0, a, b, o, n); }, 10);
        printf("2R1W plain         grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 3 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((add2_k<1>), dim3(g), dim3(256), 0, 0, a, b, o, n); }, 10);
        printf("2R1W nt-store      grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 3 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((add2_k<3>), dim3(g), dim3(256), 0, 0, a, b, o, n); }, 10);
        printf("2R1W nt-ld+st      grid %8d: %7.3f ms %7.1f GB/s\n", g, t, 3 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(g), dim3(256), 0, 0, a, (float *)o, n); }, 10);
        printf("read only          grid %8d: %7.3f ms %7.1f GB/s\n", g, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((write_k<0>), dim3(g), dim3(256), 0, 0, o, n); }, 10);
        printf("write plain        grid %8d: %7.3f ms %7.1f GB/s\n", g, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((write_k<1>), dim3(g), dim3(256), 0, 0, o, n); }, 10);
        printf("write nt           grid %8d: %7.3f ms %7.1f GB/s\n", g, t, gb / t * 1e3);
    }
    {   // one 200 KB plane per block, like the plane kernels
        const size_t per_block = 224 * 224 * 4 / 16;
        float t = time_ms([&] { hipLaunchKernelGGL((copy_blocked<0>), dim3(16384), dim3(256), 0, 0, a, o, per_block); }, 10);
        printf("copy blocked(plane) plain      : %7.3f ms %7.1f GB/s\n", t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((copy_blocked<1>), dim3(16384), dim3(256), 0, 0, a, o, per_block); }, 10);
        printf("copy blocked(plane) nt-store   : %7.3f ms %7.1f GB/s\n", t, 2 * gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((copy_blocked<3>), dim3(16384), dim3(256), 0, 0, a, o, per_block); }, 10);
        printf("copy blocked(plane) nt-ld+st   : %7.3f ms %7.1f GB/s\n", t, 2 * gb / t * 1e3);
    }
    for (size_t kb : {4, 8, 14, 28, 56, 100, 200}) {
        const size_t per_block = kb * 1024 / 16;
        const int g = (int)(n / per_block);
        for (int remap = 0; remap < 2; ++remap) {
            float t0 = time_ms([&] { hipLaunchKernelGGL((copy_blocked<0>), dim3(g), dim3(256), 0, 0, a, o, per_block); }, 10);
            float t1 = time_ms([&] { hipLaunchKernelGGL((add2_blocked<0>), dim3(g), dim3(256), 0, 0, a, b, o, per_block, remap); }, 10);
            float t2 = time_ms([&] { hipLaunchKernelGGL((add2_blocked<3>), dim3(g), dim3(256), 0, 0, a, b, o, per_block, remap); }, 10);
            printf("one-shot blocked %4zu KB/block grid %7d remap %d: copy %7.1f | 2R1W plain %7.1f | 2R1W nt %7.1f GB/s\n", kb, g, remap,
                   2 * gb / t0 * 1e3, 3 * gb / t1 * 1e3, 3 * gb / t2 * 1e3);
        }
    }
    for (size_t kb : {28, 56, 200, 800}) {
        const size_t per_block = kb * 1024 / 16;
        const int g = (int)(n / per_block);
        float t0 = time_ms([&] { hipLaunchKernelGGL((copy_pipe<0, 1>), dim3(g), dim3(256), 0, 0, a, o, per_block); }, 10);
        float t1 = time_ms([&] { hipLaunchKernelGGL((copy_pipe<0, 2>), dim3(g), dim3(256), 0, 0, a, o, per_block); }, 10);
        float t2 = time_ms([&] { hipLaunchKernelGGL((copy_pipe<0, 4>), dim3(g), dim3(256), 0, 0, a, o, per_block); }, 10);
        float t3 = time_ms([&] { hipLaunchKernelGGL((copy_pipe<3, 2>), dim3(g), dim3(256), 0, 0, a, o, per_block); }, 10);
        printf("pipelined copy %4zu KB/block grid %6d: U1 %7.1f | U2 %7.1f | U4 %7.1f | U2 nt %7.1f GB/s\n", kb, g,
               2 * gb / t0 * 1e3, 2 * gb / t1 * 1e3, 2 * gb / t2 * 1e3, 2 * gb / t3 * 1e3);
        t0 = time_ms([&] { hipLaunchKernelGGL((add2_pipe<0, 1>), dim3(g), dim3(256), 0, 0, a, b, o, per_block); }, 10);
        t1 = time_ms([&] { hipLaunchKernelGGL((add2_pipe<0, 2>), dim3(g), dim3(256), 0, 0, a, b, o, per_block); }, 10);
        t2 = time_ms([&] { hipLaunchKernelGGL((add2_pipe<3, 1>), dim3(g), dim3(256), 0, 0, a, b, o, per_block); }, 10);
        t3 = time_ms([&] { hipLaunchKernelGGL((add2_pipe<3, 2>), dim3(g), dim3(256), 0, 0, a, b, o, per_block); }, 10);
        printf("pipelined 2R1W %4zu KB/block grid %6d: U1 %7.1f | U2 %7.1f | U1 nt %7.1f | U2 nt %7.1f GB/s\n", kb, g,
               3 * gb / t0 * 1e3, 3 * gb / t1 * 1e3, 3 * gb / t2 * 1e3, 3 * gb / t3 * 1e3);
    }
#define ONESHOT(T, K) do { \
        const int g = (int)(n / ((size_t)T * K)); \
        float c0 = time_ms([&] { hipLaunchKernelGGL((copy_oneshot<T, K, 0>), dim3(g), dim3(T), 0, 0, a, o, 0); }, 10); \
        float c1 = time_ms([&] { hipLaunchKernelGGL((copy_oneshot<T, K, 3>), dim3(g), dim3(T), 0, 0, a, o, 1); }, 10); \
        float d0 = time_ms([&] { hipLaunchKernelGGL((add2_oneshot<T, K, 0>), dim3(g), dim3(T), 0, 0, a, b, o, 0); }, 10); \
        float d1 = time_ms([&] { hipLaunchKernelGGL((add2_oneshot<T, K, 3>), dim3(g), dim3(T), 0, 0, a, b, o, 0); }, 10); \
        float d2 = time_ms([&] { hipLaunchKernelGGL((add2_oneshot<T, K, 3>), dim3(g), dim3(T), 0, 0, a, b, o, 1); }, 10); \
        printf("oneshot T=%4d K=%d (%3d KB/block): copy %7.1f | copy nt+remap %7.1f | 2R1W %7.1f | 2R1W nt %7.1f | 2R1W nt+remap %7.1f GB/s\n", \
               T, K, T * K * 16 / 1024, 2 * gb / c0 * 1e3, 2 * gb / c1 * 1e3, 3 * gb / d0 * 1e3, 3 * gb / d1 * 1e3, 3 * gb / d2 * 1e3); \
    } while (0)
    ONESHOT(64, 1); ONESHOT(64, 4); ONESHOT(128, 1); ONESHOT(128, 2); ONESHOT(256, 1); ONESHOT(256, 2); ONESHOT(256, 4);
    ONESHOT(512, 1); ONESHOT(512, 2); ONESHOT(1024, 1); ONESHOT(1024, 2);
    for (int g : {1024, 2048, 4096, 8192}) {
        float t0 = time_ms([&] { hipLaunchKernelGGL((copy_gs_pipe<0>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        float t1 = time_ms([&] { hipLaunchKernelGGL((copy_gs_pipe<3>), dim3(g), dim3(256), 0, 0, a, o, n); }, 10);
        printf("grid-stride pipelined copy grid %5d: plain %7.1f | nt %7.1f GB/s\n", g, 2 * gb / t0 * 1e3, 2 * gb / t1 * 1e3);
    }
    float t = time_ms([&] { CHECK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 5);
    printf("hipMemcpy D2D                  : %7.3f ms %7.1f GB/s\n", t, 2 * gb / t * 1e3);
    return 0;
}
