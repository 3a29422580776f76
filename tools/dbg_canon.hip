#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../activesparseshifts-pytorch_amd/csrc/shiftnd_common.hpp"
using namespace shiftnd;
__global__ void k(const float* w, int len, int pad, int* out) {
    int64_t s = static_cast<int64_t>(rintf(w[threadIdx.x]));
    out[threadIdx.x*2] = canon_shift(s, len, pad);
    out[threadIdx.x*2+1] = (int)s;
}
int main(){
    float hw[4] = {9.25f, -1.5f, 2.5f, -11.75f}; float* dw; int* dout; int hout[8];
    hipMalloc(&dw, 16); hipMalloc(&dout, 32); hipMemcpy(dw, hw, 16, hipMemcpyHostToDevice);
    for (int pad = 0; pad < 5; ++pad) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dw, 7, pad, dout);
        hipMemcpy(hout, dout, 32, hipMemcpyDeviceToHost);
        printf("pad %d:", pad);
        for (int i = 0; i < 4; ++i) printf("  s=%d dev=%d host=%d", hout[2*i+1], hout[2*i], canon_shift(hout[2*i+1], 7, pad));
        printf("\n");
    }
    return 0;
}
