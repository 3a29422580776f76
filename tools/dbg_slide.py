import sys, os, subprocess
sys.path.insert(0, "/root/repo/activesparseshifts-pytorch_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from torchshifts import abi
SHAPES = [(2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (1, 3, 1, 5, 8), (1, 2, 6, 1, 32), (1, 1, 3, 37, 512),
    (3, 5, 9, 24), (2, 3, 40, 224), (5, 2, 33, 64), (1, 2, 300, 8), (2, 3, 1, 16), (7, 2, 6, 56), (1, 2, 7, 1000), (2, 1, 5, 1024)]
PLANS = [(0, 16), (100000, 2), (64, 5)]
if len(sys.argv) > 1:
    si, pi = int(sys.argv[1]), int(sys.argv[2])
    shape, plan = SHAPES[si], PLANS[pi]
    abi.set_tuning(12, 3); abi.set_tuning(13, plan[0]); abi.set_tuning(14, plan[1])
    nd = len(shape) - 2
    x = torch.rand(shape, device="cuda"); go = torch.rand(shape, device="cuda"); w = (torch.rand(shape[1], nd, device="cuda") - 0.5) * 9
    for pad in range(5):
        for active in (0, 1):
            if active:
                abi.forward(x, w, pad, active); torch.cuda.synchronize()
            abi.backward(go, w, x, pad, active); torch.cuda.synchronize()
            print("ok", pad, active, abi.last_kernel(), flush=True)
else:
    for si in range(len(SHAPES)):
        for pi in range(len(PLANS)):
            r = subprocess.run([sys.executable, __file__, str(si), str(pi)], capture_output=True, text=True)
            if r.returncode != 0:
                print("FAIL", SHAPES[si], PLANS[pi], r.stdout[-300:], r.stderr[-600:])
    print("done")
