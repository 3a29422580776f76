import os, sys, torch
sys.path.insert(0, "/root/repo/activesparseshifts-pytorch_amd")
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "activesparseshifts-pytorch_amd"))
from torchshifts import abi
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
shape = (8, 128, 16, 112, 112)
for dt in (torch.bfloat16, torch.float32):
    x = torch.rand(shape, device="cuda").to(dt)
    w = ((torch.rand(128, 3, device="cuda") * 2 - 1) * 3).to(dt)
    for cut in (None, [[1, 1], [1, 1], [1, 1]], [[0, 0], [1, 1], [1, 1]]):
        b, new = abi.check_borders(list(shape), cut, 3) if cut else (None, list(shape))
        g = torch.rand(new, device="cuda").to(dt)
        for active in (0, 1):
            out = torch.empty(new, dtype=dt, device="cuda")
            tf = timeit(lambda: abi.forward(x, w, 0, active, b, out=out)); kf = abi.last_kernel()
            gx, gw = torch.empty_like(x), torch.empty_like(w)
            ws = abi.backward_workspace(x, 0, active, b)
            tb = timeit(lambda: abi.backward(g, w, x, 0, active, b, grad_x=gx, grad_w=gw, workspace=ws)); kb = abi.last_kernel()
            es = x.element_size()
            print(str(dt)[6:], "cut", cut, "active", active, "fwd %-26s %.4f ms %.2f TB/s | bwd %-26s %.4f ms %.2f TB/s" % (kf, tf, es*(x.numel()+out.numel())/tf/1e9, kb, tb, es*(2*x.numel()+g.numel())/tb/1e9))
