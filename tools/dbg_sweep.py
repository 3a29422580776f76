import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/activesparseshifts-pytorch_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from cases import float_cases
from torchshifts import abi
n=0
for key, nd, dt, pad, active, crop, x, w, go_full, out_r, gx_r, gw_r in float_cases("g1_float.npz"):
    if active: continue
    b,_ = abi.check_borders(list(x.shape), crop, nd)
    out = abi.forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), pad, active, b).cpu().numpy()
    if not np.array_equal(out, out_r):
        bad = np.argwhere(out != out_r)
        print("MISMATCH", key, "path", abi.last_path(), "nbad", len(bad), "first", bad[:5].tolist(), "w", w[bad[0][1]].tolist())
        print(" got", out[tuple(bad[0])], "ref", out_r[tuple(bad[0])])
        n+=1
        if n>8: break
print("done", n)
