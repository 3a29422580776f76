#!/usr/bin/env python3
"""Small-tensor (C1: N2 C16 32x32 fp32) latency of the dispatcher ops: host time per call and GPU time per call,
eager and replayed from a HIP graph (the reference spends this regime in 24-byte H2D copies
and 3-4 micro-launches per call; here: no copies, 1 launch forward, 2 backward)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
import torch  # noqa: E402
import torchshifts  # noqa: E402,F401

OPS = torch.ops.torchshifts
dev = "cuda:0"
x = torch.rand(2, 16, 32, 32, device=dev)
w = torch.rand(16, 2, device=dev) * 4 - 2
go = torch.rand_like(x)
b = torch.tensor([0, 32, 0, 32, 0, 1], dtype=torch.int32)


def fwd():
    return OPS._shift2d_forward(x, w, b, [2, 16, 32, 32], 0, False)


def bwd():
    return OPS._shift2d_backward(go, w, x, b, 0, False)


def public():
    return OPS.shift2d(x, w, torch.Tensor(), 0, False)


for name, fn in (("_shift2d_forward", fwd), ("shift2d (composite: check_borders + forward)", public), ("_shift2d_backward", bwd)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / n * 1e6
    print("%-48s host %.1f us/call   end-to-end %.1f us/call" % (name, t_host, t_total))

# graph replay
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        fwd(); bwd()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        o = fwd()
        gx, gw = bwd()
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print("%-48s %.1f us per forward+backward pair" % ("HIP graph replay", (time.perf_counter() - t0) / n * 1e6))
