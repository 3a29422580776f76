#!/usr/bin/env python3
"""A fixed set of calls outside bench.py's workloads, for profiles/collect_tool.sh: channels-last tensors (N16 C256 224x224
fp32 forward / active forward / backward, quint8 C4 in NHWC), ragged rows (N128 C1024 14x14, N8 C64 225x225) and the
one-byte row kernel (N64 C256 224x224 uint8); round 3: the channels-last backward with an NCHW gradient, periodic padding through the
tiled kernels, the fused quantized shift + average pool.  Every call runs `--iters` times."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
sys.path.insert(0, ROOT)
from torchshifts import abi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="cl3d: the NDHWC forward calls of round 4; clcrop: only the cropped channels-last calls of round 4 (the same kernel instantiations "
                    "as the un-cropped ones: a run of their own keeps the per-kernel averages apart)")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    cl = torch.channels_last
    calls = []
    if a.only == "cl3d":   # NDHWC: N8 C128 16x112x112 (config 3's tensor) int32 quantized keeping the format, fp32 / bf16 sparse to NDHWC
        cl3 = torch.channels_last_3d
        shape = (8, 128, 16, 112, 112)
        xq = torch.randint(-1000, 1000, shape, dtype=torch.int32, device=dev).contiguous(memory_format=cl3)
        wq = (torch.rand(128, 3, device=dev) * 2 - 1).round().add(128).to(torch.uint8)
        oq = torch.empty_like(xq)
        xf = torch.rand(shape, device=dev).contiguous(memory_format=cl3)
        xb = xf.bfloat16()
        wf = torch.rand(128, 3, device=dev) * 2 - 1
        of, ob = torch.empty_like(xf), torch.empty_like(xb)
        calls = [lambda: abi.forward_quantized(xq, wq, 128, 3, 0, out=oq), lambda: abi.forward(xf, wf, 0, 0, out=of),
                 lambda: abi.forward(xb, wf.bfloat16(), 0, 0, out=ob), lambda: abi.forward(xf, wf, 0, 1, out=of)]
        for f in calls:
            for _ in range(a.iters):
                f()
            print(abi.last_kernel())
        torch.cuda.synchronize()
        print("done")
        return
    if a.only == "clcrop":   # N16 C256 224x224 fp32 channels-last, cut [[1, 1], [1, 1]] -> 222x222 (modules/shifts.py:41-46)
        x = torch.rand(16, 256, 224, 224, device=dev).contiguous(memory_format=cl)
        w = torch.rand(256, 2, device=dev) * 6 - 3
        b, new = abi.check_borders([16, 256, 224, 224], [[1, 1], [1, 1]], 2)
        go = torch.rand(new, device=dev).contiguous(memory_format=cl)
        go_n = torch.rand(new, device=dev)
        out_c, out_n = torch.empty_like(go), torch.empty_like(go_n)
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        ws = abi.backward_workspace(x, 0, 1, b)
        calls = [lambda: abi.forward(x, w, 0, 0, b, out=out_n), lambda: abi.forward(x, w, 0, 0, b, out=out_c), lambda: abi.forward(x, w, 0, 1, b, out=out_c),
                 lambda: abi.backward(go, w, x, 0, 0, b, grad_x=gx, grad_w=gw, workspace=ws), lambda: abi.backward(go, w, x, 0, 1, b, grad_x=gx, grad_w=gw, workspace=ws),
                 lambda: abi.backward(go_n, w, x, 0, 0, b, grad_x=gx, grad_w=gw, workspace=ws)]
        for f in calls:
            for _ in range(a.iters):
                f()
            print(abi.last_kernel())
        torch.cuda.synchronize()
        print("done")
        return
    x = torch.rand(16, 256, 224, 224, device=dev).contiguous(memory_format=cl)
    go = torch.rand(16, 256, 224, 224, device=dev).contiguous(memory_format=cl)
    w = torch.rand(256, 2, device=dev) * 6 - 3
    out_n = torch.empty(16, 256, 224, 224, device=dev)
    out_c, gx, gw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
    ws = abi.backward_workspace(x, 0, 1)
    calls += [lambda: abi.forward(x, w, 0, 0, out=out_n), lambda: abi.forward(x, w, 0, 0, out=out_c), lambda: abi.forward(x, w, 0, 1, out=out_c),
              lambda: abi.backward(go, w, x, 0, 0, grad_x=gx, grad_w=gw, workspace=ws), lambda: abi.backward(go, w, x, 0, 1, grad_x=gx, grad_w=gw, workspace=ws)]
    xq = torch.randint(0, 255, (128, 512, 56, 56), dtype=torch.uint8, device=dev).contiguous(memory_format=cl)
    wq = (torch.rand(512, 2, device=dev) * 6 - 3).round().add(128).to(torch.uint8)
    oq = torch.empty_like(xq)
    calls.append(lambda: abi.forward_quantized(xq, wq, 128, 0, 0, out=oq))
    for shape in ((128, 1024, 14, 14), (8, 64, 225, 225)):
        xs, gs = torch.rand(shape, device=dev), torch.rand(shape, device=dev)
        w2 = torch.rand(shape[1], 2, device=dev) * 6 - 3
        os_, gxs, gws = torch.empty_like(xs), torch.empty_like(xs), torch.empty_like(w2)
        wss = abi.backward_workspace(xs, 0, 1)
        calls += [lambda xs=xs, w2=w2, os_=os_: abi.forward(xs, w2, 0, 1, out=os_),
                  lambda xs=xs, gs=gs, w2=w2, gxs=gxs, gws=gws, wss=wss: abi.backward(gs, w2, xs, 0, 1, grad_x=gxs, grad_w=gws, workspace=wss)]
    xb = torch.randint(0, 255, (64, 256, 224, 224), dtype=torch.uint8, device=dev)
    wb = (torch.rand(256, 2, device=dev) * 6 - 3).round().add(128).to(torch.uint8)
    ob = torch.empty_like(xb)
    calls.append(lambda: abi.forward_quantized(xb, wb, 128, 0, 0, out=ob))
    # round 3: the mixed-layout backward (NHWC saved input, NCHW gradient), periodic padding through the tiled kernels, the fused
    # quantized shift + average pool (C4's tensor, pool 2)
    go_n = torch.rand(16, 256, 224, 224, device=dev)
    calls += [lambda: abi.backward(go_n, w, x, 0, 0, grad_x=gx, grad_w=gw, workspace=ws), lambda: abi.backward(go_n, w, x, 0, 1, grad_x=gx, grad_w=gw, workspace=ws),
              lambda: abi.forward(x, w, 2, 0, out=out_c), lambda: abi.backward(go, w, x, 2, 0, grad_x=gx, grad_w=gw, workspace=ws)]
    ob2 = abi.forward_quantized_pooled(xb, wb, 128, 3, 0, 2)   # round 4: planes beyond the plane kernel -> qpool_band_forward
    calls.append(lambda: abi.forward_quantized_pooled(xb, wb, 128, 3, 0, 2, out=ob2))
    xq2 = torch.randint(0, 255, (128, 512, 56, 56), dtype=torch.uint8, device=dev)
    oq2 = abi.forward_quantized_pooled(xq2, wq, 128, 3, 0, 2)
    calls.append(lambda: abi.forward_quantized_pooled(xq2, wq, 128, 3, 0, 2, out=oq2))
    # the 3-D walk kernels outside bench.py's C3: sparse shift, pooled forward / backward (N8 C128 16x112x112 bf16)
    x3 = torch.rand(8, 128, 16, 112, 112, device=dev).bfloat16()
    g3 = torch.rand_like(x3)
    w3 = (torch.rand(128, 3, device=dev) * 6 - 3).bfloat16()
    gx3, gw3 = torch.empty_like(x3), torch.empty_like(w3)
    ws3 = abi.backward_workspace(x3, 0, 0)
    gp3 = torch.rand(abi.pooled_shape(x3, 2), device=dev).bfloat16()
    op3 = abi.forward_pooled(x3, w3, 0, 1, 2)
    calls += [lambda: abi.backward(g3, w3, x3, 0, 0, grad_x=gx3, grad_w=gw3, workspace=ws3),
              lambda: abi.forward_pooled(x3, w3, 0, 1, 2, out=op3),
              lambda: abi.backward_pooled(gp3, w3, x3, 0, 1, 2, grad_x=gx3, grad_w=gw3)]
    for f in calls:
        for _ in range(a.iters):
            f()
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
