"""cl_tiled forward / active forward / backward timing on N16 C256 224x224 fp32 channels-last (SHIFTND_HIP_LIB=variants/<x>.so for A/B)"""
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
cl = torch.channels_last
x = torch.rand(16, 256, 224, 224, device=dev).contiguous(memory_format=cl)
go = torch.rand(16, 256, 224, 224, device=dev).contiguous(memory_format=cl)
go_n = torch.rand(16, 256, 224, 224, device=dev)
span = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
w = torch.rand(256, 2, device=dev) * 2 * span - span
out_c, gx, gw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
ws = abi.backward_workspace(x, 0, 1)
r = []
r.append(("fwd", ev(lambda: abi.forward(x, w, 0, 0, out=out_c))))
r.append(("afwd", ev(lambda: abi.forward(x, w, 0, 1, out=out_c))))
r.append(("bwd", ev(lambda: abi.backward(go, w, x, 0, 0, grad_x=gx, grad_w=gw, workspace=ws))))
r.append(("abwd", ev(lambda: abi.backward(go, w, x, 0, 1, grad_x=gx, grad_w=gw, workspace=ws))))
r.append(("bwd_nchw", ev(lambda: abi.backward(go_n, w, x, 0, 0, grad_x=gx, grad_w=gw, workspace=ws))))
print("  ".join("%s %.3f" % kv for kv in r))
