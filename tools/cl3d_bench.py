#!/usr/bin/env python3
"""NDHWC (channels_last_3d) calls through the C ABI, HIP-event times (GPU box):
    python3 tools/cl3d_bench.py [--iters 20]
N8 C128 16x112x112 (BASELINE config 3's tensor) in fp32 and bf16: the direct kernels of shiftnd_cl_tiled.hip (ND3) against the
route they replace -- shiftnd_transpose to NCDHW + the contiguous kernels (knob 20 = 0 turns the tiled kernels off)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--wrange", type=float, default=1.0, help="weights uniform in (-wrange, wrange); bench.py uses 3")
    ap.add_argument("--pad", type=int, default=0)
    a = ap.parse_args()
    dev = "cuda:0"
    cl3 = torch.channels_last_3d
    torch.manual_seed(0)
    shape = (8, 128, 16, 112, 112)
    for tdt in (torch.float32, torch.bfloat16):
        x = torch.rand(shape, device=dev).to(tdt).contiguous(memory_format=cl3)
        xn = x.contiguous()
        go_n = torch.rand(shape, device=dev).to(tdt)
        go_c = go_n.contiguous(memory_format=cl3)
        w = ((torch.rand(128, 3, device=dev) * 2 - 1) * a.wrange).to(tdt)
        es = x.element_size()
        gb = x.numel() * es / 1e9
        out_n, out_c = torch.empty_like(xn), torch.empty_like(x)
        gx_c, gx_n, gw = torch.empty_like(x), torch.empty_like(xn), torch.empty_like(w)
        for active in (0, 1):
            rows = []
            for name, fn in (("fwd -> NCDHW", lambda: abi.forward(x, w, a.pad, active, out=out_n)),
                             ("fwd -> NDHWC", lambda: abi.forward(x, w, a.pad, active, out=out_c))):
                t = ev(fn, a.iters)
                rows.append((name, abi.last_kernel(), t, 2 * gb / t))
            t = ev(lambda: abi.forward(abi.to_contiguous(x), w, a.pad, active, out=out_n), a.iters)
            rows.append(("fwd: transpose + contiguous", abi.last_kernel(), t, 2 * gb / t))
            ws = abi.backward_workspace(x, a.pad, active)
            for name, go in (("bwd, NDHWC grad", go_c), ("bwd, NCDHW grad", go_n)):
                try:
                    t = ev(lambda: abi.backward(go, w, x, a.pad, active, grad_x=gx_c, grad_w=gw, workspace=ws), a.iters)
                    rows.append((name, abi.last_kernel(), t, 3 * gb / t))
                except Exception as e:  # noqa: BLE001
                    rows.append((name, "error " + str(e)[:40], 0.0, 0.0))
            wsn = abi.backward_workspace(xn, a.pad, active)
            t = ev(lambda: abi.backward(go_n, w, abi.to_contiguous(x), a.pad, active, grad_x=gx_n, grad_w=gw, workspace=wsn), a.iters)
            rows.append(("bwd: transpose x + contiguous", abi.last_kernel(), t, 3 * gb / t))

            def op_route():   # what the op did before round 5: x to NCDHW, the contiguous kernel, grad_x back to NDHWC
                abi.backward(go_n, w, abi.to_contiguous(x), a.pad, active, grad_x=gx_n, grad_w=gw, workspace=wsn)
                return abi.to_channels_last(gx_n)
            t = ev(op_route, a.iters)
            rows.append(("bwd: transposes both ways", "", t, 3 * gb / t))
            for name, k, t, r in rows:
                print("%-8s active=%d %-32s %-28s %8.3f ms  %6.2f TB/s (algorithmic)" % (str(tdt).split(".")[-1], active, name, k, t, r))
    print("done")


if __name__ == "__main__":
    main()
