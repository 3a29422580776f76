#!/usr/bin/env python3
"""A training step of Shift3d on a dense NDHWC (channels_last_3d) input through the PUBLIC op and autograd -- the path a model takes --
against the private ops called one by one (bench.py's cl3d step): round 6's node keeps the contiguous copy of the forward.  GPU box."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "activesparseshifts-pytorch_amd"))
import torchshifts  # noqa: F401,E402
OPS = torch.ops.torchshifts


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


shape = (8, 128, 16, 112, 112)
for dt in (torch.float32, torch.bfloat16):
    x = torch.rand(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last_3d)
    w = ((torch.rand(128, 3, device="cuda") * 2 - 1) * 3).to(dt)
    b = torch.tensor([0, 16, 0, 112, 0, 112], dtype=torch.int32)
    for active in (False, True):
        g_nc = torch.rand(shape, device="cuda").to(dt)                       # what the op downstream of this forward hands back
        g_cl = g_nc.contiguous(memory_format=torch.channels_last_3d)

        def private(g):
            OPS._shift3d_forward(x, w, b, list(shape), 0, active)
            OPS._shift3d_backward(g, w, x, b, 0, active)

        def public(g):
            xr, wr = x.detach().requires_grad_(True), w.detach().requires_grad_(True)
            # (autograd.grad: a leaf's AccumulateGrad would re-lay the contiguous grad_x out in the leaf's NDHWC strides with ATen's generic
            #  copy, 1 ms -- an activation inside a network is not a leaf)
            torch.autograd.grad(OPS.shift3d(xr, wr, torch.Tensor(), 0, active), [xr, wr], g)

        print("%-9s active %d  private ops: NCDHW grad %.3f ms, NDHWC grad %.3f ms | public op + autograd: NCDHW grad %.3f ms, NDHWC grad %.3f ms" % (
            str(dt)[6:], active, ev(lambda: private(g_nc)), ev(lambda: private(g_cl)), ev(lambda: public(g_nc)), ev(lambda: public(g_cl))))
