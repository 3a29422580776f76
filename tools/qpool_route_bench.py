#!/usr/bin/env python3
"""The quantized shift + 2 x 2 average pool on the planes a quantized network pools (56 x 56 ... 14 x 14), plane kernel vs band kernel
(knob 36: 0 automatic, 2 plane kernel first, 3 qpool_band_fast first).  GPU box:  python3 tools/qpool_route_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402


def timeit(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    ev[1].synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


for shape, cut in (((128, 512, 56, 56), [[1, 1], [1, 1]]), ((128, 512, 56, 56), None), ((128, 1024, 28, 28), None), ((128, 1024, 28, 28), [[1, 1], [1, 1]]),
                   ((256, 2048, 14, 14), None), ((64, 256, 112, 112), None), ((64, 256, 112, 112), [[1, 1], [1, 1]])):
    x = (torch.rand(shape, device="cuda") * 255).to(torch.uint8)
    w = torch.randint(125, 132, (shape[1], 2), device="cuda", dtype=torch.uint8)
    b = None if cut is None else abi.check_borders(list(shape), cut, 2)[0]
    pshape = abi.pooled_shape(x, 2, b)
    out = torch.empty(pshape, dtype=torch.uint8, device="cuda")
    nbytes = x.numel() + out.numel()
    line = "%-22s cut %-18s" % (list(shape), cut)
    ref = None
    for knob in (2, 3, 0):
        abi.set_tuning(36, knob)
        ms = timeit(lambda: abi.forward_quantized_pooled(x, w, 128, 3, 0, 2, borders=b, out=out))
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref)
        line += "  knob %d %-20s %.4f ms %5.2f TB/s" % (knob, abi.last_kernel(), ms, nbytes / ms / 1e9)
    abi.set_tuning(36, 0)
    print(line)
