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
shape, dt = (64,512,224,224), torch.float16
x = torch.rand(shape, device="cuda").to(dt); w = (torch.rand(shape[1], 2, device="cuda") * 6 - 3).to(dt)
gp = torch.rand(abi.pooled_shape(x, 2), device="cuda").to(dt)
go = torch.rand(shape, device="cuda").to(dt)
gx, gw = torch.empty_like(x), torch.empty_like(w)
for knobs in ({}, {5: 2}, {4: 3}, {4: 3, 5: 2}, {3: 1}):
    for k, v in knobs.items(): abi.set_tuning(k, v)
    ws = abi.backward_workspace(x, 0, 0)
    t = ev(lambda: abi.backward_pooled(gp, w, x, 0, 0, 2, grad_x=gx, grad_w=gw, workspace=ws))
    name = abi.last_kernel()
    t2 = ev(lambda: abi.backward(go, w, x, 0, 0, grad_x=gx, grad_w=gw, workspace=ws))
    print(knobs, "pooled %.3f ms (%s)   plain %.3f ms (%s)" % (t, name, t2, abi.last_kernel()))
    for k in knobs: abi.set_tuning(k, {5: 0, 4: 1, 3: 2}[k])
