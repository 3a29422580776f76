#!/usr/bin/env python3
"""One problem through the C ABI with untouched knobs: kernel names, HIP-event times and algorithmic TB/s (GPU box).
    python3 tools/shape_bench.py 8,128,16,28,28:bf16 4,64,4,62,62:f32 ... [--pads 0,3] [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402

TDT = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}


def ev(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("problems", nargs="+")
ap.add_argument("--pads", default="0,3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--knobs", default="", help="knob=value,... (shiftnd_set_tuning)")
a = ap.parse_args()
for kv in filter(None, a.knobs.split(",")):
    abi.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
for spec in a.problems:
    shp, dt = spec.split(":")
    shape = tuple(int(v) for v in shp.split(","))
    tdt = TDT[dt]
    nd = len(shape) - 2
    x = torch.rand(shape, device="cuda").to(tdt)
    go = torch.rand(shape, device="cuda").to(tdt)
    w = (torch.rand(shape[1], nd, device="cuda") * 4 - 2).to(tdt)
    es = x.element_size()
    out, gx, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
    ws = abi.backward_workspace(x, 0, 1)
    for pad in [int(v) for v in a.pads.split(",")]:
        row = "%-24s pad %d" % (spec, pad)
        for active in (0, 1):
            tf = ev(lambda: abi.forward(x, w, pad, active, out=out), a.iters)
            kf = abi.last_kernel()
            tb = ev(lambda: abi.backward(go, w, x, pad, active, grad_x=gx, grad_w=gw, workspace=ws), a.iters)
            kb = abi.last_kernel()
            fb, bb = 2 * x.numel() * es, 3 * x.numel() * es
            row += "  | a%d %s %.4f ms %.2f TB/s  %s %.4f ms %.2f TB/s" % (active, kf, tf, fb / tf / 1e9, kb, tb, bb / tb / 1e9)
        print(row, flush=True)
