"""quantized shift + 2x2 average pool on planes beyond the plane kernel: N64 C256 224x224 uint8 (HIP-event times; knob 36 = 1: the
element-per-thread kernel)"""
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
for shape, pool in (((64, 256, 224, 224), 2), ((64, 64, 224, 224), 3), ((32, 128, 112, 112), 2)):
    x = torch.randint(0, 255, shape, dtype=torch.uint8, device=dev)
    w = (torch.rand(shape[1], 2, device=dev) * 6 - 3).round().add(128).to(torch.uint8)
    out = abi.forward_quantized_pooled(x, w, 128, 3, 0, pool)
    t = ev(lambda: abi.forward_quantized_pooled(x, w, 128, 3, 0, pool, out=out)); k = abi.last_kernel()
    abi.set_tuning(36, 1)
    t1 = ev(lambda: abi.forward_quantized_pooled(x, w, 128, 3, 0, pool, out=out)); k1 = abi.last_kernel()
    abi.set_tuning(36, 0)
    gb = (x.numel() + out.numel()) / 1e9
    print("%s pool %d: %s %.3f ms (%.2f TB/s)   %s %.3f ms (%.2f TB/s)" % (shape, pool, k, t, gb / t, k1, t1, gb / t1))
