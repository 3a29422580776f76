#!/usr/bin/env python3
"""Kernel-level A/B timing on the GPU box (HIP events, interleaved rounds in one process).

    python tools/kbench.py [--workload c2] [--rounds 5] [--iters 10] [--knobs "0=2048,4096;2=1,2,4,8"]

Times torch's own device copy (the practical HBM ceiling on this box), the forward and the backward
kernel through the C ABI, for every combination given by --knobs (knob=value lists, see
shiftnd_set_tuning).  Prints min/median ms and algorithmic GB/s.
"""
import argparse
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CUTS, WORKLOADS, synth_tensor  # noqa: E402
from torchshifts import abi  # noqa: E402


def ev_time(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pad", type=int, default=0)
    ap.add_argument("--knobs", default="")
    ap.add_argument("--wrange", type=float, default=3.0)
    ap.add_argument("--policy", type=int, default=0)
    ap.add_argument("--shape", default="", help="N,C,spatial... with --dtype / --active: a free-form workload")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--active", type=int, default=0)
    ap.add_argument("--cut", default="", help="cut amounts per dim, e.g. '1,1;1,1' (default: the workload's, bench.py CUTS)")
    ap.add_argument("--libs", default="", help="comma-separated variants/<name>.so builds timed interleaved in THIS process "
                                               "(same tensors, same physical pages); 'tree' = the in-tree library")
    a = ap.parse_args()
    extra = {  # tuning-only variants of the bench workloads
        "c2a": (2, (64, 256, 224, 224), "float32", True, "Shift2d active N64 C256 224x224 fp32"),
        "c5a": (2, (64, 512, 224, 224), "float16", True, "Shift2d active N64 C512 224x224 fp16"),
        "c3f": (3, (8, 128, 16, 112, 112), "float32", True, "Shift3d active N8 C128 16x112x112 fp32"),
        "c3s": (3, (8, 128, 16, 112, 112), "bfloat16", False, "Shift3d SSL N8 C128 16x112x112 bf16"),
        "c3fs": (3, (8, 128, 16, 112, 112), "float32", False, "Shift3d SSL N8 C128 16x112x112 fp32"),
        "r56": (2, (128, 512, 56, 56), "float32", False, "Shift2d SSL N128 C512 56x56 fp32"),
        "c2n32": (2, (32, 256, 224, 224), "float32", False, "Shift2d SSL N32 C256 224x224 fp32"),
        "c2n128": (2, (128, 256, 224, 224), "float32", False, "Shift2d SSL N128 C256 224x224 fp32"),
        "c2c64": (2, (64, 64, 224, 224), "float32", False, "Shift2d SSL N64 C64 224x224 fp32"),
        "c2s448": (2, (16, 256, 448, 448), "float32", False, "Shift2d SSL N16 C256 448x448 fp32"),
        "c2s160": (2, (64, 256, 160, 160), "float32", False, "Shift2d SSL N64 C256 160x160 fp32"),
        "d1": (1, (256, 512, 4096), "float32", False, "Shift1d SSL N256 C512 L4096 fp32"),
        "d1a": (1, (256, 512, 4096), "float32", True, "Shift1d active N256 C512 L4096 fp32"),
        "d1h": (1, (256, 512, 4096), "float16", False, "Shift1d SSL N256 C512 L4096 fp16"),
    }
    if a.shape:
        shp = tuple(int(v) for v in a.shape.split(","))
        extra["free"] = (len(shp) - 2, shp, a.dtype, bool(a.active), "Shift%dd %s %s %s" % (len(shp) - 2, "active" if a.active else "SSL", a.shape, a.dtype))
        a.workload = "free"
    nd, shape, dtname, active, desc = {**WORKLOADS, **extra}[a.workload]
    dev = torch.device("cuda:0")
    abi.set_path_policy(a.policy)
    quant = dtname == "quint8"
    dtype = torch.float32 if quant else getattr(torch, dtname)
    cuts = CUTS.get(a.workload)
    if a.cut:
        cuts = [[int(v) for v in part.split(",")] for part in a.cut.split(";")]
    borders, oshape = abi.check_borders(list(shape), cuts, nd) if cuts else (None, list(shape))   # (cropped windows: bench.py's CUTS)
    x = synth_tensor(torch, shape, 1, dev, dtype)
    go = synth_tensor(torch, tuple(oshape), 2, dev, dtype)
    w = synth_tensor(torch, (shape[1], nd), 3, dev, torch.float32, -a.wrange, a.wrange).to(dtype)
    if quant:
        x = (x * 255).to(torch.uint8)
        wq = (w.float().round() + 128).to(torch.uint8)
    out, gx, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
    elems, oelems, es = x.numel(), go.numel(), x.element_size()
    knobs = []
    for part in [p for p in a.knobs.split(";") if p]:
        k, vals = part.split("=")
        knobs.append((int(k), [int(v) for v in vals.split(",")]))
    combos = list(itertools.product(*[[(k, v) for v in vals] for k, vals in knobs])) or [()]
    libs = {}
    for name in [n for n in a.libs.split(",") if n]:
        abi._LIB_PATH = (os.path.join(ROOT, "activesparseshifts-pytorch_amd", "torchshifts", "libshiftnd_hip.so") if name == "tree"
                         else os.path.join(ROOT, "variants", name + ".so"))
        abi._lib = None
        libs[name] = abi.lib()
    combos = [(ln, c) for c in combos for ln in (libs or {"": None})]
    res, kernels = {}, {}
    for r in range(a.rounds + 1):
        for ci, (ln, combo) in enumerate(combos):
            if ln:
                abi._lib = libs[ln]
                abi.set_path_policy(a.policy)
            for k, v in combo:
                abi.set_tuning(k, v)
            ws = None if quant else abi.backward_workspace(x, a.pad, active, borders)
            tag = (ln + ":" if ln else "") + (",".join("%d=%d" % kv for kv in combo) or "default")
            if quant:
                fns = {"fwd[" + tag + "]": lambda: abi.forward_quantized(x, wq, 128, 0, a.pad, out=out)}
            else:
                fns = {"fwd[" + tag + "]": lambda: abi.forward(x, w, a.pad, active, borders, out=out),
                       "bwd[" + tag + "]": lambda: abi.backward(go, w, x, a.pad, active, borders, grad_x=gx, grad_w=gw, workspace=ws)}
            if ci == 0:
                fns["copy"] = lambda: gx.copy_(x)
            for name, fn in fns.items():
                fn()
                kernels.setdefault(name, abi.last_kernel() if name != "copy" else "torch copy")
                torch.cuda.synchronize()
                t = ev_time(fn, a.iters)
                if r > 0:
                    res.setdefault(name, []).append(t)
    print("workload:", desc, "pad", a.pad)
    print("kernels:", kernels)
    for name, ts in res.items():
        nbytes = {"c": 2 * elems, "f": elems + oelems, "b": 2 * elems + oelems}[name[0]] * es
        print("%-40s min %8.3f ms  med %8.3f ms  %8.1f GB/s (min)" % (name, min(ts), statistics.median(ts),
                                                                       nbytes / min(ts) / 1e6))


if __name__ == "__main__":
    main()
