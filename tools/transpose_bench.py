#!/usr/bin/env python3
"""shiftnd_transpose (channels-last <-> contiguous) against torch's layout copies: correctness and time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
import torch  # noqa: E402
from torchshifts import abi  # noqa: E402


def ev(fn, it=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it


for shape, dt in [((16, 256, 224, 224), torch.float32), ((64, 512, 224, 224), torch.float16), ((128, 512, 56, 56), torch.uint8),
                  ((8, 128, 16, 112, 112), torch.bfloat16), ((3, 70, 13, 9), torch.float32), ((2, 5, 7, 3), torch.int8)]:
    x = (torch.rand(shape, device="cuda") * 100).to(dt)
    fmt = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    xc = x.contiguous(memory_format=fmt)
    a = abi.to_contiguous(xc)
    b = abi.to_channels_last(x)
    assert a.is_contiguous() and torch.equal(a, x) and torch.equal(b, x) and b.stride() == xc.stride()
    t1, t2 = ev(lambda: abi.to_contiguous(xc)), ev(lambda: abi.to_channels_last(x))
    t3, t4 = ev(lambda: xc.contiguous()), ev(lambda: x.contiguous(memory_format=fmt))
    gb = 2 * x.numel() * x.element_size() / 1e6
    print("%-26s %-14s CL->NCHW %.3f ms (%.0f GB/s; torch %.3f)   NCHW->CL %.3f ms (%.0f GB/s; torch %.3f)"
          % (shape, str(dt).split(".")[1], t1, gb / t1, t3, t2, gb / t2, t4))
