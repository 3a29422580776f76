// LDS access-pattern micro-benchmark (gfx950): cycles per wave-instruction per CU for the patterns the staged kernels
// use -- 16-byte-per-lane rows (lane stride 16 B) read / written aligned, misaligned by a dword, or dword by dword.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_bench.hip -o tools/lds_bench && tools/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, int disp) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = i;
    __syncthreads();
    // lane l of the workgroup: row = l / 14, chunk = l % 14 (224-byte rows, pitch given by KIND >> 4)
    const int pitch = (KIND & 0x100) ? 256 : 288;
    const int row = threadIdx.x / 14, tc = threadIdx.x % 14;
    const uint32_t base = row * pitch + 32 + tc * 16 + disp * 4;
    uint32_t acc = 0;
    u4 v = {1u, 2u, 3u, 4u};
    for (int it = 0; it < iters; ++it) {
        const uint32_t a = base + (it & 7) * 4608;  // (stay inside 32 KiB; defeat hoisting)
        switch (KIND & 0xff) {
        case 0: { u4 q; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a)); acc += q.x; break; }
        case 1: { uint32_t q; asm volatile("ds_read_b32 %0, %1 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a)); acc += q; break; }
        case 2: { asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v)); break; }
        case 3: { asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n ds_write2_b32 %0, %3, %4 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); break; }
        case 4: { u4 q, r; asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(q), "=v"(r) : "v"(a)); acc += q.x + r.y; break; }
        case 5: { typedef uint32_t u2 __attribute__((ext_vector_type(2))); u2 q; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a)); acc += q.x; break; }
        }
    }
    if (acc == 0x12345) out[0] = acc;
}

template <int KIND> void run(const char *name, int disp, int instr) {
    uint32_t *out;
    (void)hipMalloc(&out, 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(256), 0, 0, out, 10, disp);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(256), 0, 0, out, iters, disp);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 4 workgroups x 4 waves x iters x instr wave-instructions
    printf("%-44s disp %d: %7.3f ms  %6.1f ns per wave-instruction per CU\n", name, disp, ms, ms * 1e6 / (16.0 * iters * instr));
    (void)hipFree(out);
}

int main() {
    for (int d = 0; d < 2; ++d) {
        run<0>("ds_read_b128, 14 lanes/row, pitch 288", d, 1);
        run<0x100>("ds_read_b128, 14 lanes/row, pitch 256", d, 1);
        run<1>("ds_read_b32 (+16), lane stride 16 B, pitch 288", d, 1);
        run<0x101>("ds_read_b32 (+16), lane stride 16 B, pitch 256", d, 1);
        run<2>("ds_write_b128, pitch 288", d, 1);
        run<0x102>("ds_write_b128, pitch 256", d, 1);
        run<3>("2 x ds_write2_b32 (one piece), pitch 288", d, 2);
        run<0x103>("2 x ds_write2_b32 (one piece), pitch 256", d, 2);
        run<4>("2 x ds_read_b128 (32 B per lane), pitch 288", d, 2);
        run<0x104>("2 x ds_read_b128 (32 B per lane), pitch 256", d, 2);
        run<5>("ds_read_b64, pitch 288", d, 1);
    }
    return 0;
}
