#!/usr/bin/env python3
"""Does a build of the library show the round-4 race of walk_backward16<.., ZEROS = false>?  (GPU box; SHIFTND_HIP_LIB=variants/<x>.so)
C3's tensor (N8 C128 16x112x112 bf16), sparse and interpolating shift, border / reflect padding: grad_x of `--reps` runs against
the first one and against a run with one workgroup per CU's worth of work (N1 C8: never raced).  Prints the number of differing runs."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(0)
shape = (8, 128, 16, 112, 112)
x = torch.rand(shape, device="cuda").to(torch.bfloat16)
go = torch.rand(shape, device="cuda").to(torch.bfloat16)
w = (torch.rand(128, 3, device="cuda") * 6 - 3).to(torch.bfloat16)
bad_total = 0
for pad in (1, 3):
    for active in (0, 1):
        # reference: the same channels in slices of 8 with N = 1 (112 workgroups: at most one per CU)
        ref = torch.empty_like(x)
        for n in range(shape[0]):
            for c0 in range(0, 128, 8):
                gx, _ = abi.backward(go[n:n + 1, c0:c0 + 8].contiguous(), w[c0:c0 + 8].contiguous(), x[n:n + 1, c0:c0 + 8].contiguous(), pad, active)
                ref[n, c0:c0 + 8] = gx[0]
        bad = 0
        zeros = 0
        for r in range(a.reps):
            gx, _ = abi.backward(go, w, x, pad, active)
            kern = abi.last_kernel()
            if not torch.equal(gx, ref):
                bad += 1
                d = gx != ref
                zeros += int(((gx == 0) & d).sum().item())
                first = d.nonzero()[0].tolist()
        print("pad %d active %d %s: %d of %d runs differ%s" % (pad, active, kern, bad, a.reps, (", %d wrong zeros, first at %s" % (zeros, first)) if bad else ""))
        bad_total += bad
print("RACE" if bad_total else "clean")
