"""cl_tiled_forward timing: N16 C256 224x224 fp32 channels-last input, NCHW and channels-last output, band-rows sweep.
   SHIFTND_HIP_LIB=variants/<x>.so python3 tools/cl_tiled_bench.py [band_rows ...]"""
import sys
import torch
sys.path.insert(0, "activesparseshifts-pytorch_amd"); sys.path.insert(0, ".")
from torchshifts import abi


def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(it): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best


dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(16, 256, 224, 224, device=dev); w = torch.rand(256, 2, device=dev) * 6 - 3
xc = x.contiguous(memory_format=torch.channels_last)
o = torch.empty_like(x); ocl = torch.empty_like(xc); ref = torch.empty_like(x)
abi.forward(x, w, 0, 0, out=ref)
for br in [int(a) for a in sys.argv[1:]] or [0]:
    abi.set_tuning(21, br)
    t1 = ev(lambda: abi.forward(xc, w, 0, 0, out=o)); k1 = abi.last_kernel()
    t2 = ev(lambda: abi.forward(xc, w, 0, 0, out=ocl)); k2 = abi.last_kernel()
    assert torch.equal(o, ref) and torch.equal(ocl, ref)
    print("band_rows %3d  CL->NCHW %.3f ms (%s, %.0f GB/s)   CL->CL %.3f ms (%s)" % (br, t1, k1, 8 * x.numel() / t1 / 1e6, t2, k2))
t0 = ev(lambda: abi.forward(x, w, 0, 0, out=o))
print("NCHW->NCHW %.3f ms (%s)" % (t0, abi.last_kernel()))
