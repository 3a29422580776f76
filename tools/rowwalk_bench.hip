// rowwalk_bench.hip -- does the HBM care how long the contiguous pieces of a row walk are?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 tools/rowwalk_bench.hip -o tools/rowwalk_bench && tools/rowwalk_bench
// The access pattern of the sliding-window kernels on C3 (bf16 N8 C128 16x112x112: rows of 224 bytes): a workgroup
// owns the 16 planes of one (n, c) volume, a thread one 16-byte chunk column of one plane; per step every thread moves
// R consecutive rows (R x 224 contiguous bytes per plane) from the input to the output, loads one step ahead.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int R, bool STORE>
__global__ __launch_bounds__(256) void rowwalk(const char *__restrict__ x, char *__restrict__ o, int planes, int rows, int rb, int bands) {
    const int cpr = rb / 16;
    const int vol = blockIdx.x / bands, band = blockIdx.x % bands;
    const int s = threadIdx.x / cpr, tc = threadIdx.x % cpr;
    if (s >= planes) return;
    const int seg = rows / bands, r0 = band * seg;
    const size_t base = ((size_t)vol * planes + s) * rows * rb + (size_t)r0 * rb + tc * 16;
    u4 v[R], acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = *(const u4 *)(x + base + (size_t)j * rb);
    for (int t = 0; t < seg; t += R) {
        u4 cur[R];
#pragma unroll
        for (int j = 0; j < R; ++j) cur[j] = v[j];
        const int tn = t + R < seg ? t + R : t;
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = *(const u4 *)(x + base + (size_t)(tn + j) * rb);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (STORE) *(u4 *)(o + base + (size_t)(t + j) * rb) = cur[j];
            else acc += cur[j];
        }
    }
    if (!STORE && acc.x == 0x12345678) *(u4 *)(o + base) = acc;
}

// loads one row per step, stores deferred: R rows kept in registers and written together every R steps
template <int R>
__global__ __launch_bounds__(256) void rowwalk_defer(const char *__restrict__ x, char *__restrict__ o, int planes, int rows, int rb, int bands) {
    const int cpr = rb / 16;
    const int vol = blockIdx.x / bands, band = blockIdx.x % bands;
    const int s = threadIdx.x / cpr, tc = threadIdx.x % cpr;
    if (s >= planes) return;
    const int seg = rows / bands, r0 = band * seg;
    const size_t base = ((size_t)vol * planes + s) * rows * rb + (size_t)r0 * rb + tc * 16;
    u4 v = *(const u4 *)(x + base);
    for (int t = 0; t < seg; t += R) {
        u4 keep[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            keep[j] = v;
            const int tn = t + j + 1 < seg ? t + j + 1 : t + j;
            v = *(const u4 *)(x + base + (size_t)tn * rb);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) *(u4 *)(o + base + (size_t)(t + j) * rb) = keep[j];
    }
}
template <int R> float run_defer(const char *x, char *o, int vols, int planes, int rows, int rb, int bands) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rowwalk_defer<R>), dim3(vols * bands), dim3(256), 0, 0, x, o, planes, rows, rb, bands);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    return best;
}

template <int R, bool STORE> float run(const char *x, char *o, int vols, int planes, int rows, int rb, int bands) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rowwalk<R, STORE>), dim3(vols * bands), dim3(256), 0, 0, x, o, planes, rows, rb, bands);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    return best;
}

int main() {
    const int vols = 1024, planes = 16, rows = 112, rb = 224;
    const size_t bytes = (size_t)vols * planes * rows * rb;
    char *x, *o;
    CHECK(hipMalloc(&x, bytes)); CHECK(hipMalloc(&o, bytes));
    CHECK(hipMemset(x, 1, bytes)); CHECK(hipMemset(o, 0, bytes));
    printf("C3-shaped volume walk, %.0f MB per tensor; rows of %d bytes\n", bytes / 1e6, rb);
    for (int bands = 1; bands <= 4; bands *= 2) {
        printf("bands %d (workgroups %d)\n", bands, vols * bands);
#define LINE(R) { const float c = run<R, true>(x, o, vols, planes, rows, rb, bands), r = run<R, false>(x, o, vols, planes, rows, rb, bands); \
        printf("  %d row(s) per step: copy %.3f ms %.0f GB/s   read-only %.3f ms %.0f GB/s\n", R, c, 2 * bytes / c / 1e6, r, bytes / r / 1e6); }
        LINE(1) LINE(2) LINE(4) LINE(7)
#define DLINE(R) { const float c = run_defer<R>(x, o, vols, planes, rows, rb, bands); printf("  1 row loaded per step, %d rows stored together: copy %.3f ms %.0f GB/s\n", R, c, 2 * bytes / c / 1e6); }
        DLINE(2) DLINE(4) DLINE(7)
    }
    return 0;
}
