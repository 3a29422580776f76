#!/usr/bin/env python3
"""The flat-stream kernels (csrc/shiftnd_flat.hip) through the C ABI, HIP-event times per shape (GPU box):
    python3 tools/flat_bench.py [--iters 20] [--old]      (--old: knob 27 = 1, the kernels they replace)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402


def ev(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--old", action="store_true")
ap.add_argument("--only", default=None)
ap.add_argument("--pads", default="0,3")
a = ap.parse_args()
if a.old:
    abi.set_tuning(27, 1)
SHAPES = [("r14", (128, 1024, 14, 14), torch.float32, None), ("r7", (128, 2048, 7, 7), torch.float32, None), ("r14h", (128, 1024, 14, 14), torch.float16, None),
          ("r28h", (128, 512, 28, 28), torch.bfloat16, [[1, 1], [1, 1]]),
          ("r62", (512, 16, 62, 62), torch.float32, None), ("r62c", (512, 16, 62, 62), torch.float32, [[1, 1], [1, 1]]), ("r113", (16, 64, 113, 113), torch.float32, None),
          ("r222", (64, 256, 222, 222), torch.float32, None), ("r225", (8, 64, 225, 225), torch.float32, None), ("r222h", (64, 256, 222, 222), torch.float16, None),
          ("c2crop", (64, 256, 224, 224), torch.float32, [[1, 1], [1, 1]]), ("c1d", (32, 512, 4096), torch.float32, None)]
for name, shape, tdt, cut in SHAPES:
    if a.only and name not in a.only.split(","):
        continue
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), cut, nd) if cut else (None, list(shape))
    x = torch.rand(shape, device="cuda").to(tdt)
    go = torch.rand(new, device="cuda").to(tdt)
    w = (torch.rand(shape[1], nd, device="cuda") * 6 - 3).to(tdt)
    es = x.element_size()
    out, gx, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
    ws = abi.backward_workspace(x, 0, 0, b)
    for pad in [int(v) for v in a.pads.split(",")]:
        row = "%-6s pad %d" % (name, pad)
        for active in (0, 1):
            tf = ev(lambda: abi.forward(x, w, pad, active, b, out=out), a.iters)
            kf = abi.last_kernel()
            tb = ev(lambda: abi.backward(go, w, x, pad, active, b, grad_x=gx, grad_w=gw, workspace=ws), a.iters)
            kb = abi.last_kernel()
            fb, bb = (x.numel() + go.numel()) * es, (2 * x.numel() + go.numel()) * es
            row += "  | a%d %s %.4f ms %.2f  %s %.4f ms %.2f" % (active, kf[:16], tf, fb / tf / 8e9, kb[:16], tb, bb / tb / 8e9)
        print(row)
