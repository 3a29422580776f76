// stream_probe.hip -- same-box calibration of the HBM streams the shiftnd kernels are made of (diagnostic binary,
// built by build(); bench.py runs it as a child process BEFORE it touches the GPU and reports
// roofline.box_stream / frac_of_box next to the 8 TB/s fraction).
//
//   tools/stream_probe                 one JSON line: best plain 1R1W (copy) and 2R1W (read two tensors, write one)
//                                      stream rates on C2-sized buffers (3.29 GB each), float4 accesses
//   tools/stream_probe --explore       the experiments behind DESIGN section 9 (cache-policy bits, window spread,
//                                      relative stream offsets, bytes in flight)
//
// Every kernel here is a plain stream: no index maps, no shifts, no reduction.  "Best" = the maximum over a small
// fixed set of launch shapes (one-shot blocks of 256 threads x K float4, nontemporal or plain), each timed over
// several launches with HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, static_cast<int>(bytes), 0x00020000);
}

__device__ __forceinline__ size_t block_of(int perm_groups) {
    // perm_groups G: consecutive dispatch ids are spread over G far-apart regions (G = 1: linear sweep front)
    size_t blk = blockIdx.x;
    if (perm_groups > 1) {
        const size_t per = gridDim.x / perm_groups;
        blk = (blockIdx.x % perm_groups) * per + blockIdx.x / perm_groups;
    }
    return blk;
}

// one-shot stream: a block of 256 threads moves 256*K float4; NR = number of tensors read (1 or 2); LP / SP = aux bits
// of the raw-buffer loads / stores (bit0 sc0, bit1 nt, bit4 sc1)
template <int K, int NR, int LP, int SP>
__global__ __launch_bounds__(256) void stream_oneshot(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o,
                                                      int perm_groups, int lds_pad) {
    extern __shared__ char pad_[];
    if (lds_pad < 0) pad_[threadIdx.x] = 0;  // keeps the dynamic LDS allocation (occupancy limiter) alive
    const size_t blk = block_of(perm_groups);
    const size_t base = blk * static_cast<size_t>(256 * K);  // wave-uniform resource base, per-lane offset
    const int lane_off = threadIdx.x * 16;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(a + base, 0x7ffffffcu), rb = make_rsrc(b + base, 0x7ffffffcu), ro = make_rsrc(o + base, 0x7ffffffcu);
    f4 x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        x[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, lane_off + k * 4096, 0, LP));
        if (NR == 2) y[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rb, lane_off + k * 4096, 0, LP));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        f4 r = NR == 2 ? x[k] + y[k] : x[k];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, r), ro, lane_off + k * 4096, 0, SP);
    }
}

// read-only / write-only one-shot
template <int K, int LP>
__global__ __launch_bounds__(256) void read_oneshot(const f4 *__restrict__ a, float *__restrict__ sink) {
    const size_t base = blockIdx.x * static_cast<size_t>(256 * K);
    const int lane_off = threadIdx.x * 16;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(a + base, 0x7ffffffcu);
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; ++k) acc += __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, lane_off + k * 4096, 0, LP));
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}
template <int K, int SP>
__global__ __launch_bounds__(256) void write_oneshot(f4 *__restrict__ o) {
    const size_t base = blockIdx.x * static_cast<size_t>(256 * K);
    const int lane_off = threadIdx.x * 16;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + base, 0x7ffffffcu);
    const f4 v = {1, 2, 3, 4};
#pragma unroll
    for (int k = 0; k < K; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), ro, lane_off + k * 4096, 0, SP);
}

static hipEvent_t g_e0, g_e1;
template <typename F> static double time_ms(F f, int iters, int reps = 3) {
    f();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(g_e0, 0));
        for (int i = 0; i < iters; ++i) f();
        CHECK(hipEventRecord(g_e1, 0));
        CHECK(hipEventSynchronize(g_e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, g_e0, g_e1));
        if (ms / iters < best) best = ms / iters;
    }
    return best;
}


// ---- mock of the LDS-staged backward's traffic (X6): a step = 5 source rows of a + 4 rows of b by LDS-DMA into one
// tile (504 pieces of 16 bytes), barrier, 4 output rows (224 pieces) computed from LDS and stored.  The step order is
// the experiment: a workgroup owns `run` consecutive steps, then jumps ahead by (workgroups of its XCD) x run steps
// (persistent linear sweep: 8 fronts, one per XCD), or -- persistent = 0 -- it owns ONE run and exits (run = 14: the
// band walk of plane_backward_lds).
// one-shot single-step mock with ROWS output rows per workgroup and, DEP = 1, a dependent scalar table load in front of
// the DMA (the per-channel descriptor a real kernel needs before it knows its source rows)
template <int ROWS, int LP, int SP, int DEP, int T>
__global__ __launch_bounds__(T) void mock_step(const char *__restrict__ a, const char *__restrict__ b, char *__restrict__ o,
                                                 const int *__restrict__ table, unsigned steps_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    constexpr unsigned NA = (ROWS + 1) * 56, NB = ROWS * 56, NP = NA + NB, NO = ROWS * 56, SB = ROWS * 896;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    if (j >= steps_per_xcd) return;
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const unsigned step = xcd * steps_per_xcd + j;
    size_t base = static_cast<size_t>(step) * SB;
    if (DEP) {
        const unsigned plane = step / (224 / ROWS);
        const int d = table[plane & 255];  // uniform: s_load; zero, but the address below depends on it
        base += static_cast<size_t>(d) * 896;
    }
#pragma unroll
    for (unsigned k = 0; k < (NP + T - 1) / T; ++k) {
        const unsigned q = k * T + tid;
        if (q < NP) {
            const char *src = q < NA ? a + base + q * 16 : b + base + (q - NA) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(tile + (k * T + wave * 64) * 16), 16, 0, LP);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < (NO + T - 1) / T; ++k) {
        const unsigned q = k * T + tid;
        if (q < NO) {
            const f4 *t = reinterpret_cast<const f4 *>(tile);
            f4 r, r2, r3;
            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + q))));
            asm volatile("ds_read_b128 %0, %1" : "=v"(r2) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + q + 56))));
            asm volatile("ds_read_b128 %0, %1" : "=v"(r3) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + q + NA))));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const f4 res = r2 - r + r3;
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + static_cast<size_t>(step) * SB, SB);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, res), ro, q * 16, 0, SP);
        }
    }
}

template <int ROWS, int LP, int SP, int DEP, int T>
static double run_step(const char *a, const char *b, char *o, const int *table, size_t bytes) {
    const unsigned steps_per_xcd = static_cast<unsigned>(bytes / (ROWS * 896)) / 8;
    const size_t lds = ((ROWS + 1) * 56 + ROWS * 56 + T - 1) / T * T * 16;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&mock_step<ROWS, LP, SP, DEP, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const double t = time_ms([&] { hipLaunchKernelGGL((mock_step<ROWS, LP, SP, DEP, T>), dim3(steps_per_xcd * 8), dim3(T), lds, 0, a, b, o, table, steps_per_xcd); }, 5);
    return static_cast<double>(steps_per_xcd) * 8 * ROWS * 896 * 3 / t / 1e6;
}

template <int TILES, int LP, int SP>
__global__ __launch_bounds__(256) void mock_backward(const char *__restrict__ a, const char *__restrict__ b, char *__restrict__ o,
                                                     unsigned total_steps, unsigned run, int persistent) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const unsigned steps_per_xcd = total_steps / 8;
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    unsigned run0 = j * run;                      // first step of this workgroup's current run, relative to its XCD's range
    const unsigned stride = per_xcd * run;
    unsigned buf = 0;
    bool primed = false;
    f4 acc = {0, 0, 0, 0};
    auto issue = [&](unsigned step, char *dst) {
        const size_t base = static_cast<size_t>(xcd * steps_per_xcd + step) * 3584;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned q = k * 256 + tid;
            if (q < 504) {
                const char *src = q < 280 ? a + base + q * 16 : b + base + (q - 280) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(dst + (k * 256 + wave * 64) * 16), 16, 0, LP);
            }
        }
    };
    while (run0 < steps_per_xcd) {
        const unsigned nrun = min(run, steps_per_xcd - run0);
        if (TILES == 2 && !primed) { issue(run0, tile + buf * 8192); primed = true; }
        for (unsigned l = 0; l < nrun; ++l) {
            const unsigned step = run0 + l;
            char *cur = tile + (TILES == 2 ? buf * 8192 : 0);
            if (TILES == 1) issue(step, cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (TILES == 2) {
                unsigned nxt = step + 1;
                bool have = l + 1 < nrun;
                if (!have && persistent && run0 + stride < steps_per_xcd) { nxt = run0 + stride; have = true; }
                if (have) issue(nxt, tile + (buf ^ 1) * 8192);
            }
            if (tid < 224) {
                const f4 *t = reinterpret_cast<const f4 *>(cur);
                f4 r;
                asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + tid))));
                f4 r2, r3;
                asm volatile("ds_read_b128 %0, %1" : "=v"(r2) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + tid + 56))));
                asm volatile("ds_read_b128 %0, %1" : "=v"(r3) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(t + tid + 280))));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += r * r3;
                const f4 res = r2 - r + r3;
                const size_t base = static_cast<size_t>(xcd * steps_per_xcd + step) * 3584;
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + base, 3584);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, res), ro, tid * 16, 0, SP);
            }
            if (TILES == 1) __syncthreads();
            buf ^= 1;
        }
        if (!persistent) break;
        run0 += stride;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) o[0] = 1;
}

template <int TILES, int LP, int SP>
static double run_mock(const char *a, const char *b, char *o, size_t bytes, unsigned run, int persistent, int wgs_per_cu, int pad_lds = 0) {
    const unsigned total_steps = static_cast<unsigned>(bytes / 3584) / 8 * 8;
    unsigned grid;
    if (persistent) grid = 256 * wgs_per_cu;
    else grid = (total_steps / 8 + run - 1) / run * 8;
    const size_t lds = TILES * 8192 + pad_lds;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&mock_backward<TILES, LP, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const double t = time_ms([&] { hipLaunchKernelGGL((mock_backward<TILES, LP, SP>), dim3(grid), dim3(256), lds, 0, a, b, o, total_steps, run, persistent); }, 5);
    return static_cast<double>(total_steps) * 3584 * 3 / t / 1e6;
}

template <int K, int NR, int LP, int SP>
static double run_oneshot(const f4 *a, const f4 *b, f4 *o, size_t n, int perm = 1, int lds = 0, int iters = 6) {
    const int g = static_cast<int>(n / (256 * K));
    const double t = time_ms([&] { hipLaunchKernelGGL((stream_oneshot<K, NR, LP, SP>), dim3(g), dim3(256), lds, 0, a, b, o, perm, lds > 0 ? 1 : 0); }, iters);
    const double bytes = static_cast<double>(g) * 256 * K * 16 * (NR + 1);
    return bytes / t / 1e6;  // GB/s
}

int main(int argc, char **argv) {
    const bool explore = argc > 1 && !strcmp(argv[1], "--explore");
    const bool mock = argc > 1 && !strcmp(argv[1], "--mock");
    size_t bytes = 64ull * 256 * 224 * 224 * 4;  // one C2 tensor; --bytes N: the tensor size of another workload (a short
    for (int i = 1; i + 1 < argc; ++i)                 // kernel pays launch ramp and tail: its ceiling is lower than C2's)
        if (!strcmp(argv[i], "--bytes")) bytes = (strtoull(argv[i + 1], nullptr, 10) + 16383) & ~static_cast<size_t>(16383);
    const size_t n = bytes / 16;
    const size_t slack = 64ull << 20;
    char *pool;
    CHECK(hipMalloc(&pool, 3 * bytes + 4 * slack));
    CHECK(hipMemset(pool, 1, 3 * bytes + 4 * slack));
    CHECK(hipEventCreate(&g_e0));
    CHECK(hipEventCreate(&g_e1));
    const f4 *a = reinterpret_cast<const f4 *>(pool);
    const f4 *b = reinterpret_cast<const f4 *>(pool + bytes + slack);
    f4 *o = reinterpret_cast<f4 *>(pool + 2 * (bytes + slack));

    // ---- calibration line (always) -----------------------------------------------------------------------------
    double c1 = 0, c2 = 0, v;
    const char *w1 = "", *w2 = "";
#define TRY1(K, LP, SP, name) do { v = run_oneshot<K, 1, LP, SP>(a, b, o, n); if (v > c1) { c1 = v; w1 = name; } } while (0)
#define TRY2(K, LP, SP, name) do { v = run_oneshot<K, 2, LP, SP>(a, b, o, n); if (v > c2) { c2 = v; w2 = name; } } while (0)
    TRY1(1, 0, 0, "K1 plain"); TRY1(2, 0, 0, "K2 plain"); TRY1(4, 0, 0, "K4 plain");
    TRY1(1, 2, 2, "K1 nt"); TRY1(2, 2, 2, "K2 nt"); TRY1(4, 2, 2, "K4 nt"); TRY1(4, 0, 2, "K4 nt-store");
    TRY2(1, 0, 0, "K1 plain"); TRY2(2, 0, 0, "K2 plain"); TRY2(4, 0, 0, "K4 plain");
    TRY2(1, 2, 2, "K1 nt"); TRY2(2, 2, 2, "K2 nt"); TRY2(4, 2, 2, "K4 nt"); TRY2(2, 0, 2, "K2 nt-store");
    double rd = 0, wr = 0;
    {
        const int g = static_cast<int>(n / (256 * 4));
        double t = time_ms([&] { hipLaunchKernelGGL((read_oneshot<4, 0>), dim3(g), dim3(256), 0, 0, a, reinterpret_cast<float *>(o)); }, 6);
        rd = bytes / t / 1e6;
        t = time_ms([&] { hipLaunchKernelGGL((write_oneshot<4, 2>), dim3(g), dim3(256), 0, 0, o); }, 6);
        wr = bytes / t / 1e6;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"probe\": \"tools/stream_probe\", \"buffer_bytes\": %zu, \"1R1W_GBps\": %.1f, \"1R1W_shape\": \"%s\", \"2R1W_GBps\": %.1f, "
           "\"2R1W_shape\": \"%s\", \"read_GBps\": %.1f, \"write_GBps\": %.1f, \"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n",
           bytes, c1, w1, c2, w2, rd, wr, prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    fflush(stdout);
    if (mock) {
        const char *ca = reinterpret_cast<const char *>(a), *cb = reinterpret_cast<const char *>(b);
        char *co = reinterpret_cast<char *>(o);
        printf("X6 mock of the staged backward (2R1W through LDS, 8 KB per step); GB/s algorithmic\n");
        printf("  band walk (one run of 14 steps per workgroup, as plane_backward_lds): 1 tile plain %7.1f  nt %7.1f | 2 tiles plain %7.1f nt %7.1f\n",
               run_mock<1, 0, 2>(ca, cb, co, bytes, 14, 0, 0), run_mock<1, 2, 2>(ca, cb, co, bytes, 14, 0, 0),
               run_mock<2, 0, 2>(ca, cb, co, bytes, 14, 0, 0), run_mock<2, 2, 2>(ca, cb, co, bytes, 14, 0, 0));
        for (unsigned run : {56u, 28u, 7u, 4u, 2u, 1u})
            printf("  one-shot runs of %2u steps: 1 tile plain %7.1f nt %7.1f | 2 tiles nt %7.1f\n", run,
                   run_mock<1, 0, 2>(ca, cb, co, bytes, run, 0, 0), run_mock<1, 2, 2>(ca, cb, co, bytes, run, 0, 0), run_mock<2, 2, 2>(ca, cb, co, bytes, run, 0, 0));
        {
            int *table;
            CHECK(hipMalloc(&table, 1024));
            CHECK(hipMemset(table, 0, 1024));
            printf("  single-step one-shot workgroups, ROWS output rows each (threads): nt / nt + dependent table load / plain loads\n");
            printf("    ROWS 1 ( 64): %7.1f / %7.1f / %7.1f\n", run_step<1, 2, 2, 0, 64>(ca, cb, co, table, bytes), run_step<1, 2, 2, 1, 64>(ca, cb, co, table, bytes), run_step<1, 0, 2, 1, 64>(ca, cb, co, table, bytes));
            printf("    ROWS 2 (128): %7.1f / %7.1f / %7.1f\n", run_step<2, 2, 2, 0, 128>(ca, cb, co, table, bytes), run_step<2, 2, 2, 1, 128>(ca, cb, co, table, bytes), run_step<2, 0, 2, 1, 128>(ca, cb, co, table, bytes));
            printf("    ROWS 4 (256): %7.1f / %7.1f / %7.1f\n", run_step<4, 2, 2, 0, 256>(ca, cb, co, table, bytes), run_step<4, 2, 2, 1, 256>(ca, cb, co, table, bytes), run_step<4, 0, 2, 1, 256>(ca, cb, co, table, bytes));
            printf("    ROWS 8 (256): %7.1f / %7.1f / %7.1f\n", run_step<8, 2, 2, 0, 256>(ca, cb, co, table, bytes), run_step<8, 2, 2, 1, 256>(ca, cb, co, table, bytes), run_step<8, 0, 2, 1, 256>(ca, cb, co, table, bytes));
            printf("    ROWS 8 (512): %7.1f / %7.1f / %7.1f\n", run_step<8, 2, 2, 0, 512>(ca, cb, co, table, bytes), run_step<8, 2, 2, 1, 512>(ca, cb, co, table, bytes), run_step<8, 0, 2, 1, 512>(ca, cb, co, table, bytes));
            printf("    ROWS 16 (256): %7.1f / %7.1f / %7.1f\n", run_step<16, 2, 2, 0, 256>(ca, cb, co, table, bytes), run_step<16, 2, 2, 1, 256>(ca, cb, co, table, bytes), run_step<16, 0, 2, 1, 256>(ca, cb, co, table, bytes));
        }
        if (argc > 2)
        for (int wpc : {2, 3, 4, 6, 8, 12, 16})
            for (unsigned run : {1u, 2u, 4u, 14u}) {
                printf("  persistent sweep %2d wgs/CU run %2u: 1 tile plain %7.1f nt %7.1f | 2 tiles plain %7.1f nt %7.1f\n", wpc, run,
                       run_mock<1, 0, 2>(ca, cb, co, bytes, run, 1, wpc), run_mock<1, 2, 2>(ca, cb, co, bytes, run, 1, wpc),
                       run_mock<2, 0, 2>(ca, cb, co, bytes, run, 1, wpc), run_mock<2, 2, 2>(ca, cb, co, bytes, run, 1, wpc));
            }
        return 0;
    }
    if (!explore) return 0;

    // ---- X1: cache-policy bits of loads and stores -------------------------------------------------------------
    printf("X1 policy sweep (K4 one-shot; aux: 1 sc0, 2 nt, 16 sc1)\n");
#define POL(LP, SP) printf("  load aux %2d store aux %2d: copy %7.1f  2R1W %7.1f GB/s\n", LP, SP, run_oneshot<4, 1, LP, SP>(a, b, o, n), run_oneshot<4, 2, LP, SP>(a, b, o, n))
    POL(0, 0); POL(2, 0); POL(0, 2); POL(2, 2); POL(1, 1); POL(16, 16); POL(17, 17); POL(18, 18); POL(19, 19); POL(3, 3);
    POL(0, 16); POL(0, 17); POL(0, 18); POL(0, 19); POL(2, 18); POL(2, 19); POL(16, 2); POL(18, 2); POL(17, 2); POL(19, 2);

    // ---- X2: window spread ------------------------------------------------------------------------------------
    printf("X2 window spread (dispatch ids spread over G regions; K4 nt)\n");
    for (int G : {1, 2, 4, 8, 16, 64, 256, 1024, 4096, 16384}) {
        printf("  G %6d: copy %7.1f  2R1W %7.1f  | K1 copy %7.1f 2R1W %7.1f GB/s\n", G, run_oneshot<4, 1, 2, 2>(a, b, o, n, G), run_oneshot<4, 2, 2, 2>(a, b, o, n, G),
               run_oneshot<1, 1, 2, 2>(a, b, o, n, G), run_oneshot<1, 2, 2, 2>(a, b, o, n, G));
    }

    // ---- X3: relative offsets of the streams ------------------------------------------------------------------
    printf("X3 stream offsets (b and o displaced by d, 2d bytes against a's position; K4 nt and K4 plain)\n");
    for (size_t d : {0ul, 256ul, 1024ul, 4096ul, 8192ul, 16384ul, 65536ul, 262144ul, 1048576ul, 2097152ul + 4096, 4194304ul, 16777216ul + 65536, 33554432ul}) {
        const f4 *b2 = reinterpret_cast<const f4 *>(pool + bytes + slack + d);
        f4 *o2 = reinterpret_cast<f4 *>(pool + 2 * (bytes + slack) + 2 * d);
        printf("  d %9zu: 2R1W nt %7.1f plain %7.1f | copy(a->o2) nt %7.1f GB/s\n", d, run_oneshot<4, 2, 2, 2>(a, b2, o2, n), run_oneshot<4, 2, 0, 0>(a, b2, o2, n),
               run_oneshot<4, 1, 2, 2>(a, b2, o2, n));
    }

    // ---- X4: bytes in flight (occupancy limited through dynamic LDS) --------------------------------------------
    printf("X4 occupancy (dynamic LDS per block limits blocks per CU; 16 KB of loads per block at K4, 4 KB at K1)\n");
    for (int lds : {0, 16384, 20000, 26000, 32768, 40000, 53000, 65536}) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_oneshot<4, 1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_oneshot<4, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_oneshot<1, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_oneshot<8, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        printf("  lds %6d (<= %2d blocks/CU): copy K4 %7.1f  2R1W K4 %7.1f  2R1W K1 %7.1f  2R1W K8 %7.1f GB/s\n", lds, lds ? 163840 / lds : 8,
               run_oneshot<4, 1, 2, 2>(a, b, o, n, 1, lds), run_oneshot<4, 2, 2, 2>(a, b, o, n, 1, lds), run_oneshot<1, 2, 2, 2>(a, b, o, n, 1, lds),
               run_oneshot<8, 2, 2, 2>(a, b, o, n, 1, lds));
    }
    // ---- X5: read-only / write-only with policies ---------------------------------------------------------------
    {
        const int g = static_cast<int>(n / (256 * 4));
        double t;
#define RD(LP) t = time_ms([&] { hipLaunchKernelGGL((read_oneshot<4, LP>), dim3(g), dim3(256), 0, 0, a, reinterpret_cast<float *>(o)); }, 6); printf("  read aux %2d: %7.1f GB/s\n", LP, bytes / t / 1e6)
#define WR(SP) t = time_ms([&] { hipLaunchKernelGGL((write_oneshot<4, SP>), dim3(g), dim3(256), 0, 0, o); }, 6); printf("  write aux %2d: %7.1f GB/s\n", SP, bytes / t / 1e6)
        printf("X5 read-only / write-only\n");
        RD(0); RD(2); RD(16); RD(18); RD(19); WR(0); WR(2); WR(16); WR(18); WR(19);
    }
    return 0;
}
