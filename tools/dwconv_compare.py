#!/usr/bin/env python3
"""The reference's README says its CUDA shift is "still slower than PyTorch's 3x3 DW Convolution".  On the GPU box: the
shift op of this repo against torch's depthwise 3x3 convolution (MIOpen) on the same tensor, forward and backward.
    python3 tools/dwconv_compare.py [--shape 64,256,224,224] [--dtype float32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
sys.path.insert(0, ROOT)
import torchshifts  # noqa: E402,F401


def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,256,224,224")
    ap.add_argument("--dtype", default="float32")
    a = ap.parse_args()
    shape = [int(v) for v in a.shape.split(",")]
    dt = getattr(torch, a.dtype)
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.rand(shape, device=dev, dtype=dt, requires_grad=True)
    C = shape[1]
    wconv = torch.rand(C, 1, 3, 3, device=dev, dtype=dt, requires_grad=True)
    wshift = ((torch.rand(C, 2, device=dev) - 0.5) * 2).to(dt).requires_grad_(True)
    go = torch.rand(shape, device=dev, dtype=dt)
    conv = lambda: torch.nn.functional.conv2d(x, wconv, None, 1, 1, 1, C)
    for name, active in (("sparse shift", False), ("active shift", True)):
        shift = lambda: torch.ops.torchshifts.shift2d(x, wshift, torch.Tensor(), 0, active)
        tf_s = ev(lambda: shift())
        y = shift()
        tb_s = ev(lambda: torch.autograd.grad(y, (x, wshift), go, retain_graph=True))
        print("%-13s forward %.3f ms   backward %.3f ms   step %.3f ms" % (name, tf_s, tb_s, tf_s + tb_s))
    tf_c = ev(lambda: conv())
    y = conv()
    tb_c = ev(lambda: torch.autograd.grad(y, (x, wconv), go, retain_graph=True))
    print("%-13s forward %.3f ms   backward %.3f ms   step %.3f ms   (torch.nn.functional.conv2d, groups = C, 3x3, padding 1)"
          % ("depthwise 3x3", tf_c, tb_c, tf_c + tb_c))


if __name__ == "__main__":
    main()
