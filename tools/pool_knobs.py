import sys, torch
sys.path.insert(0, "activesparseshifts-pytorch_amd"); sys.path.insert(0, ".")
from torchshifts import abi
def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
for shape, dt, act in [((64,256,224,224), torch.float32, 0), ((64,256,224,224), torch.float32, 1), ((64,512,224,224), torch.float16, 0), ((128,512,56,56), torch.float32, 0)]:
    x = torch.rand(shape, device="cuda").to(dt); w = (torch.rand(shape[1], 2, device="cuda") * 6 - 3).to(dt)
    gp = torch.rand(abi.pooled_shape(x, 2), device="cuda").to(dt)
    gx, gw = torch.empty_like(x), torch.empty_like(w); ws = abi.backward_workspace(x, 0, act)
    for k7 in (0, 65536, 16384, 8192, 4096):
        abi.set_tuning(7, k7)
        ws = abi.backward_workspace(x, 0, act)
        t = ev(lambda: abi.backward_pooled(gp, w, x, 0, act, 2, grad_x=gx, grad_w=gw, workspace=ws))
        print(shape, dt, "active", act, "knob7", k7, "%.3f ms" % t, abi.last_kernel())
    abi.set_tuning(7, 0)
