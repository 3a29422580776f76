#!/usr/bin/env python3
"""Seeded fuzz of the cropped 3-D routes (round 6: the window inside the walk kernels, crop_backward3 / crop_forward3 behind them):
tools/fuzz_round2.py's case_crop3d with zeros padding every other case, plus cropped volumes behind the (K0, K1, 2) pool in fp32 / fp64
against the oracle's fused form.  GPU box:  python3 tools/fuzz_crop_walks.py --seconds 150 --seed 0"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_round2 as F  # noqa: E402
from fuzz_round2 import abi, O, DEV, rel_err, weights, _random_crop  # noqa: E402


def case_pooled(rs):
    dt = [np.float32, np.float64][rs.randint(2)]
    es = np.dtype(dt).itemsize
    per16 = 16 // es
    sp = (int(rs.choice([2, 3, 5, 8])), int(rs.choice([2, 5, 9, 18, 37])), per16 * int(rs.choice([1, 2, 3, 7, 14])))
    N, C = int(rs.randint(1, 3)), int(rs.randint(1, 4))
    shape = (N, C) + sp
    crop = _random_crop(rs, sp)
    b, new = abi.check_borders(list(shape), crop, 3)
    pool = (int(rs.choice([1, 2])), int(rs.choice([1, 2, 3])), 2)
    pad = 0 if rs.rand() < 0.6 else int(rs.randint(0, 5))
    active = int(rs.randint(0, 2))
    x = rs.uniform(-1, 1, size=shape).astype(dt)
    w = weights(rs, C, 3, sp, 3.5).astype(dt)
    xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    ref = O.forward_pooled(x, w, pad, active, pool, b)
    gp = rs.uniform(-1, 1, size=ref.shape).astype(dt)
    gx_r, gw_r = O.backward_pooled(gp, w, x, pad, active, pool, b)
    try:
        gx, gw = abi.backward_pooled(torch.from_numpy(gp).to(DEV), wd, xd, pad, active, pool, b)
    except RuntimeError as e:   # (a geometry the fused kernels do not serve: the op composes the pool's backward and the shift's)
        assert "not served" in str(e), e
        F.count["not fused"] += 1
        return
    F.count[abi.last_kernel()] += 1
    key = ("pool3", shape, crop, pool, dt.__name__, pad, active, abi.last_kernel())
    assert np.array_equal(gx.cpu().numpy(), gx_r), key
    assert rel_err(gw.cpu().numpy(), gw_r) < (1e-12 if dt == np.float64 else 2e-5), key


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    zero_pad = _ZeroPad(rs)
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        if n % 3 == 2:
            case_pooled(rs)
        else:   # (every other case under zeros padding: the routes of the cropped walks)
            F.case_crop3d(zero_pad if n % 2 == 0 else rs)
        n += 1
    print("OK %d cases in %.0f s" % (n, time.time() - t0))
    for k, v in sorted(F.count.items(), key=lambda kv: -kv[1]):
        print("  %-32s %d" % (k, v))


class _ZeroPad:
    """a RandomState whose randint(0, 5) -- case_crop3d's padding draw -- returns 0"""

    def __init__(self, rs):
        self._rs = rs

    def randint(self, *a, **k):
        v = self._rs.randint(*a, **k)
        return 0 if a == (0, 5) else v

    def __getattr__(self, name):
        return getattr(self._rs, name)


if __name__ == "__main__":
    main()
