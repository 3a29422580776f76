import sys, torch
sys.path.insert(0, "activesparseshifts-pytorch_amd"); sys.path.insert(0, ".")
import torchshifts
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
OPS = torch.ops.torchshifts
b6 = torch.tensor([0, 224, 0, 224, 0, 1], dtype=torch.int32)
cl = torch.channels_last
for tdt in (torch.bfloat16, torch.float32):
    x = torch.rand(16, 256, 224, 224, device=dev).to(tdt); go = torch.rand(16, 256, 224, 224, device=dev).to(tdt)
    w = (torch.rand(256, 2, device=dev) * 6 - 3).to(tdt)
    xc, goc = x.contiguous(memory_format=cl), go.contiguous(memory_format=cl)
    for active in (False, True):
        for tiled in (1, 0):
            abi.set_tuning(20, tiled)
            tf = ev(lambda: OPS._shift2d_forward(xc, w, b6, list(x.shape), 0, active)); kf = abi.last_kernel()
            tb = ev(lambda: OPS._shift2d_backward(goc, w, xc, b6, 0, active)); kb = abi.last_kernel()
            print("%s active=%d tiled=%d  op fwd (NHWC in) %.3f ms (%s)   op bwd (all NHWC) %.3f ms (%s)" % (str(tdt)[6:], active, tiled, tf, kf, tb, kb))
abi.set_tuning(20, 1)
