// SALU issue-rate micro-benchmark (gfx950): ns per scalar instruction per CU at 4..32 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t a) {
    uint32_t s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                         "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(a) : "scc");
        } else {
            asm volatile("s_mul_i32 %0, %0, %8\n s_mul_i32 %1, %1, %8\n s_mul_i32 %2, %2, %8\n s_mul_i32 %3, %3, %8\n"
                         "s_mul_i32 %4, %4, %8\n s_mul_i32 %5, %5, %8\n s_mul_i32 %6, %6, %8\n s_mul_i32 %7, %7, %8\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(a) : "scc");
        }
    }
    if ((s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 0x12345 && threadIdx.x == 0) out[0] = s0;
}
template <int KIND> void run(const char *name) {
    uint32_t *out; (void)hipMalloc(&out, 4);
    const int iters = 20000;
    printf("%-10s", name);
    for (int wpc : {1, 2, 4, 8}) {  // workgroups (4 waves each) per CU
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wpc), dim3(256), 0, 0, out, 10, 3u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wpc), dim3(256), 0, 0, out, iters, 3u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_cu = (double)wpc * 4 * iters * 8;  // wave-instructions per CU
        printf("  %2d waves/CU: %.3f ns/inst/CU", wpc * 4, ms * 1e6 / inst_per_cu);
    }
    printf("\n");
}
int main() { run<0>("s_add_u32"); run<1>("s_mul_i32"); return 0; }
