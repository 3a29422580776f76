#!/usr/bin/env python3
"""Fused shift + average pool vs the two-step sequence (shift kernel, then torch's avg_pool) on the GPU box.

    python tools/pool_bench.py [--shape 64,256,224,224] [--pool 2] [--dtype float32] [--active 0] [--pad 0]

Times with HIP events: forward  = shiftnd_forward + avg_pool{N}d            vs shiftnd_forward_pooled
                       backward = avg_pool backward + shiftnd_backward      vs shiftnd_backward_pooled
and prints the algorithmic traffic of each form (fused forward: read x + write pooled; fused backward:
read pooled grad + read x + write grad_x).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from torchshifts import abi  # noqa: E402


def ev_time(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def quant_main(a, shape, nd):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    xq = torch.randint(0, 256, shape, device=dev, dtype=torch.uint8)
    wq = torch.randint(125, 132, (shape[1], nd), device=dev, dtype=torch.uint8)
    y = torch.empty_like(xq)
    out = abi.forward_quantized_pooled(xq, wq, 128, 3, a.pad, a.pool)
    fns = {"qshift alone": lambda: abi.forward_quantized(xq, wq, 128, 3, a.pad, out=y),
           "qshift+pool": lambda: abi.forward_quantized_pooled(xq, wq, 128, 3, a.pad, a.pool, out=out)}
    traffic = {"qshift alone": 2 * xq.numel(), "qshift+pool": xq.numel() + out.numel()}
    best = {k: 1e9 for k in fns}
    for k, fn in fns.items():
        fn()
        print(k, "kernel:", abi.last_kernel())
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k, fn in fns.items():
            best[k] = min(best[k], ev_time(fn, a.iters))
    print("shape %s pool %d uint8 pad %d" % (shape, a.pool, a.pad))
    for k in fns:
        print("%-12s %8.4f ms   %7.1f GB/s algorithmic (%.3f GB)" % (k, best[k], traffic[k] / best[k] / 1e6, traffic[k] / 1e9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,256,224,224")
    ap.add_argument("--pool", type=int, default=2)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--active", type=int, default=0)
    ap.add_argument("--pad", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--no-step", action="store_true", help="knob 32 = 1: the band-walk kernels instead of the one-step kernels")
    ap.add_argument("--force-fused", action="store_true", help="path policy 2: fuse the backward even where the automatic choice would not")
    ap.add_argument("--knob", action="append", default=[], help="K=V tuning knob (repeatable)")
    ap.add_argument("--quant", action="store_true", help="one-byte quantized tensors: shiftnd_forward_quantized alone vs "
                    "shiftnd_forward_quantized_pooled (ATen has no QuantizedCUDA pool to compare with)")
    a = ap.parse_args()
    shape = [int(v) for v in a.shape.split(",")]
    nd = len(shape) - 2
    for kv in a.knob:
        k, v = kv.split("=")
        abi.set_tuning(int(k), int(v))
    if a.quant:
        return quant_main(a, shape, nd)
    if a.no_step:
        abi.set_tuning(32, 1)
    dt = getattr(torch, a.dtype)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.rand(shape, device=dev).to(dt)
    w = (torch.rand(shape[1], nd, device=dev) * 6 - 3).to(dt)
    pool_fn = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}[nd]
    y = abi.forward(x, w, a.pad, a.active)
    ref = pool_fn(y, a.pool, a.pool, ceil_mode=True)
    out = torch.empty_like(ref)
    gp = torch.rand_like(ref)
    gx, gw = torch.empty_like(x), torch.empty_like(w)
    ws = abi.backward_workspace(x, a.pad, a.active)
    bwd_name = {1: "avg_pool2d_backward", 2: "avg_pool2d_backward", 3: "avg_pool3d_backward"}[nd]
    pool_bwd = getattr(torch.ops.aten, bwd_name)
    k = [a.pool] * nd

    def unf_fwd():
        abi.forward(x, w, a.pad, a.active, out=y)
        return pool_fn(y, a.pool, a.pool, ceil_mode=True)

    def fus_fwd():
        abi.forward_pooled(x, w, a.pad, a.active, a.pool, out=out)

    def unf_bwd():
        if nd == 1:
            g = pool_bwd(gp.unsqueeze(2), y.unsqueeze(2), [1, a.pool], [1, a.pool], [0, 0], True, True, None).squeeze(2)
        else:
            g = pool_bwd(gp, y, k, k, [0] * nd, True, True, None)
        abi.backward(g, w, x, a.pad, a.active, grad_x=gx, grad_w=gw, workspace=ws)

    def fus_bwd():
        if a.force_fused:
            abi.set_path_policy(2)
        abi.backward_pooled(gp, w, x, a.pad, a.active, a.pool, grad_x=gx, grad_w=gw, workspace=ws)
        abi.set_path_policy(0)

    es = x.element_size()
    n, npool = x.numel(), ref.numel()
    traffic = {"fwd unfused": (2 * n + n + npool) * es, "fwd fused": (n + npool) * es,
               "bwd unfused": (npool + n + 3 * n) * es, "bwd fused": (npool + 2 * n) * es}
    fns = {"fwd unfused": unf_fwd, "fwd fused": fus_fwd, "bwd unfused": unf_bwd, "bwd fused": fus_bwd}
    best = {kk: 1e9 for kk in fns}
    for kk, fn in fns.items():
        fn()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for kk, fn in fns.items():
            best[kk] = min(best[kk], ev_time(fn, a.iters))
    print("shape %s pool %d %s active %d pad %d" % (shape, a.pool, a.dtype, a.active, a.pad))
    print("backward kernel:", abi.last_kernel())
    for kk in fns:
        print("%-12s %8.3f ms   %7.1f GB/s algorithmic (%.2f GB)" % (kk, best[kk], traffic[kk] / best[kk] / 1e6, traffic[kk] / 1e9))


if __name__ == "__main__":
    main()
