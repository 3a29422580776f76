// VALU issue-rate micro-benchmark (gfx950): cycles per wave64 VALU instruction per SIMD, at 1, 2, 4 and 8 waves per SIMD,
// for the instructions the shiftnd kernels are made of.  Each case is one inline-asm instruction applied to 16
// independent registers per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench && tools/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CASES(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
    X(1, "v_add_f32", "v_add_f32 %0, %0, %1") \
    X(2, "v_lshlrev_b32", "v_lshlrev_b32 %0, 16, %0") \
    X(3, "v_and_b32", "v_and_b32 %0, %1, %0") \
    X(4, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16") \
    X(5, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(6, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(7, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0") \
    X(8, "v_cvt_f32_bf16", "v_cvt_f32_bf16 %0, %0") \
    X(9, "v_dot2_f32_bf16", "v_dot2_f32_bf16 %0, %1, %2, %0") \
    X(10, "v_dot2_f32_f16", "v_dot2_f32_f16 %0, %1, %2, %0") \
    X(11, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2") \
    X(12, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1") \
    X(13, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
    X(14, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %0, %1") \
    X(15, "v_fma_mix_f32", "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]") \
    X(16, "v_add_f64", "v_add_f64 %0, %0, %1") \
    X(17, "v_cvt_f64_f32", "v_cvt_f64_f32 %0, %1") \
    X(18, "v_sub_f32", "v_sub_f32 %0, %0, %1") \
    X(19, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
    X(20, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(21, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1") \
    X(22, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 9") \
    X(23, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
    X(24, "v_lshrrev_b32", "v_lshrrev_b32 %0, 16, %0") \
    X(25, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1") \
    X(26, "v_dot2c_f32_bf16", "v_dot2c_f32_bf16 %0, %1, %2") \
    X(27, "v_dot2c_f32_f16", "v_dot2c_f32_f16 %0, %1, %2") \
    X(28, "v_lshlrev_b32 (vgpr amount)", "v_lshlrev_b32 %0, %1, %0") \
    X(29, "v_mov_b32_sdwa word0->word1", "v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0") \
    X(30, "v_cndmask_b32 e64 sgpr mask", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]") \
    X(31, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
    X(32, "v_or_b32", "v_or_b32 %0, %0, %1") \
    X(33, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 16, %1") \
    X(34, "v_max_f32", "v_max_f32 %0, %0, %1") \
    X(35, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[16];
    f2 pa = {a, a}, pb = {b, b};
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#define X(ID, NAME, ASM) \
            if (KIND == ID) { \
                if (ID == 11 || ID == 12 || ID == 16 || ID == 25) asm volatile(ASM : "+v"(v[i]) : "v"(pa), "v"(pb)); \
                else if (ID == 17) asm volatile(ASM : "=v"(v[i]) : "v"(pa.x), "v"(pb.x)); \
                else asm volatile(ASM : "+v"(v[i].x) : "v"(pa.x), "v"(pb.x)); \
            }
            CASES(X)
#undef X
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND> void run(const char *name) {
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 10000;
    printf("%-20s", name);
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)wps * iters * 16;
        printf("  %dw: %5.2f ns", wps, ms * 1e6 / inst);
    }
    printf("   (per instruction per SIMD)\n");
    (void)hipFree(out);
}

int main() {
#define X(ID, NAME, ASM) run<ID>(NAME);
    CASES(X)
#undef X
    return 0;
}
