#!/usr/bin/env python3
"""Where the module configurations stand: Shift{1,2,3}d x {sparse, interpolating} x {no cut, emulate_dw cut 1/1} x {no pool, stride-2
pool} x {fp32, bf16}, forward and backward through the dispatcher ops on contiguous tensors; algorithmic TB/s (forward x + out, backward
grad + x + grad_x) and the kernels that ran.  GPU box:  python3 tools/module_matrix.py [--small]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
import torchshifts  # noqa: F401,E402
from torchshifts import abi  # noqa: E402

ops = torch.ops.torchshifts


def timeit(fn, iters=10):
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
ap.add_argument("--small", action="store_true")
ap.add_argument("--pad", type=int, default=0, help="padding mode 0..4 (zeros, border, periodic, reflect, symmetric)")
ap.add_argument("--dtypes", default="float32,bfloat16")
ap.add_argument("--channels-last", action="store_true", help="2-D / 3-D inputs (and gradients of the unpooled ops) as dense channels-last tensors")
a = ap.parse_args()
PAD = a.pad
SHAPES = {1: (256, 512, 4096), 2: (32, 256, 112, 112), 3: (8, 128, 16, 56, 56)}
if a.small:
    SHAPES = {1: (64, 128, 1024), 2: (8, 64, 56, 56), 3: (4, 32, 8, 28, 28)}
rows = []
for nd, shape in SHAPES.items():
    for dt in [getattr(torch, n) for n in a.dtypes.split(",")]:
        if a.channels_last and nd == 1:
            continue
        x = torch.rand(shape, device="cuda").to(dt)
        if a.channels_last:
            x = x.contiguous(memory_format=torch.channels_last if nd == 2 else torch.channels_last_3d)
        w = ((torch.rand(shape[1], nd, device="cuda") * 2 - 1) * 2.5).to(dt)
        es = x.element_size()
        for cut in (None, [[1, 1]] * nd):
            b, new = abi.check_borders(list(shape), cut, nd) if cut else (abi.default_borders(x), list(shape))
            bt = torch.tensor(b, dtype=torch.int32)
            for pool in (None, 2):
                for active in (False, True):
                    if pool:
                        fop, bop = getattr(ops, "_shift%dd_pool_forward" % nd), getattr(ops, "_shift%dd_pool_backward" % nd)
                        out = fop(x, w, bt, new, [pool] * nd, PAD, active)
                        g = torch.rand_like(out)
                        tf = timeit(lambda: fop(x, w, bt, new, [pool] * nd, PAD, active))
                        kf = abi.last_kernel()
                        tb = timeit(lambda: bop(g, w, x, bt, [pool] * nd, PAD, active))
                        kb = abi.last_kernel()
                    else:
                        fop, bop = getattr(ops, "_shift%dd_forward" % nd), getattr(ops, "_shift%dd_backward" % nd)
                        out = fop(x, w, bt, new, PAD, active)
                        g = torch.rand_like(out)
                        tf = timeit(lambda: fop(x, w, bt, new, PAD, active))
                        kf = abi.last_kernel()
                        tb = timeit(lambda: bop(g, w, x, bt, PAD, active))
                        kb = abi.last_kernel()
                    fb, bb = es * (x.numel() + out.numel()), es * (2 * x.numel() + out.numel())
                    rows.append((fb / tf / 1e9, "fwd", nd, str(dt)[6:], cut is not None, pool, active, kf, tf))
                    rows.append((bb / tb / 1e9, "bwd", nd, str(dt)[6:], cut is not None, pool, active, kb, tb))
rows.sort()
print("%-4s %-3s %-9s %-5s %-5s %-6s %-34s %9s %7s" % ("dir", "nd", "dtype", "cut", "pool", "active", "last kernel", "ms", "TB/s"))
for r in rows:
    print("%-4s %-3d %-9s %-5s %-5s %-6s %-34s %9.4f %7.2f" % (r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[0]))
