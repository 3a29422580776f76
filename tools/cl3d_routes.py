#!/usr/bin/env python3
"""NDHWC (channels_last_3d) Shift3d: the direct kernels against the layout-change route, piece by piece (GPU box).
    python3 tools/cl3d_routes.py [--shape 8,128,16,112,112] [--dtype float32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="8,128,16,112,112")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--wrange", type=float, default=3.0)
a = ap.parse_args()
shape = [int(v) for v in a.shape.split(",")]
nd = len(shape) - 2
dt = getattr(torch, a.dtype)
fmt = torch.channels_last_3d if nd == 3 else torch.channels_last
x = torch.rand(shape, device="cuda").to(dt).contiguous(memory_format=fmt)
g = torch.rand(shape, device="cuda").to(dt).contiguous(memory_format=fmt)
w = ((torch.rand(shape[1], nd, device="cuda") * 2 - 1) * a.wrange).to(dt)
xc, gc = x.contiguous(), g.contiguous()
nbytes = x.numel() * x.element_size()
print("shape", shape, a.dtype, "tensor %.1f MB" % (nbytes / 1e6))
t = timeit(lambda: abi.to_contiguous(x))
print("transpose CL -> contiguous      %.4f ms  %.2f TB/s (1R1W)" % (t, 2 * nbytes / t / 1e9))
t = timeit(lambda: abi.to_channels_last(xc))
print("transpose contiguous -> CL      %.4f ms  %.2f TB/s (1R1W)" % (t, 2 * nbytes / t / 1e9))
for active in (0, 1):
    out = torch.empty(shape, dtype=dt, device="cuda")
    t1 = timeit(lambda: abi.forward(x, w, 0, active, out=out))
    k1 = abi.last_kernel()
    t2 = timeit(lambda: abi.forward(xc, w, 0, active, out=out))
    k2 = abi.last_kernel()
    print("active %d forward  direct %-28s %.4f ms | contiguous %-24s %.4f ms" % (active, k1, t1, k2, t2))
    gx_cl, gx_c, gw = torch.empty_like(x), torch.empty_like(xc), torch.empty_like(w)
    ws = abi.backward_workspace(x, 0, active)
    t1 = timeit(lambda: abi.backward(g, w, x, 0, active, grad_x=gx_cl, grad_w=gw, workspace=ws))
    k1 = abi.last_kernel()
    t1b = timeit(lambda: abi.backward(gc, w, x, 0, active, grad_x=gx_cl, grad_w=gw, workspace=ws))
    k1b = abi.last_kernel()
    wsc = abi.backward_workspace(xc, 0, active)
    t2 = timeit(lambda: abi.backward(gc, w, xc, 0, active, grad_x=gx_c, grad_w=gw, workspace=wsc))
    k2 = abi.last_kernel()
    print("active %d backward direct %-28s %.4f ms | NCDHW grad %-32s %.4f ms | contiguous %-22s %.4f ms" % (active, k1, t1, k1b, t1b, k2, t2))
    ops = torch.ops.torchshifts
    b = torch.tensor(abi.default_borders(x), dtype=torch.int32)
    fop, bop = getattr(ops, "_shift%dd_forward" % nd), getattr(ops, "_shift%dd_backward" % nd)
    tf = timeit(lambda: fop(x, w, b, shape, 0, bool(active)))
    tb = timeit(lambda: bop(g, w, x, b, 0, bool(active)))
    tb2 = timeit(lambda: bop(gc, w, x, b, 0, bool(active)))
    print("active %d through the op: forward %.4f ms, backward (NDHWC grad) %.4f ms, backward (NCDHW grad) %.4f ms" % (active, tf, tb, tb2))
