#!/usr/bin/env python3
"""Time the REAL reference CPU kernels (oracle/_ref/_C.so) -- bench.py's cpu_baseline leg.

TEST/BENCH INFRASTRUCTURE ONLY.  Run as a child process (it registers the same `torchshifts::*`
op names as the product library, so the two cannot share a process).  Prints one JSON line.
Falls back to the plain-C oracle port (oracle/shift_oracle.c) when oracle/_ref is not present.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def synth(shape, seed):
    """the repo's integer-hash stream (bench.py: synth_tensor), numpy version"""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    h = (idx * np.uint64(2654435761) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return ((h >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24)).reshape(shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--c", type=int, default=256)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--pad", type=int, default=0)
    a = ap.parse_args()
    shape = (a.n, a.c, a.hw, a.hw)
    x = synth(shape, 1)
    go = synth(shape, 2)
    w = (synth((a.c, 2), 3) * 6 - 3).astype(np.float32)
    elems = int(np.prod(shape))
    ref_so = os.path.join(HERE, "_ref", "_C.so")
    if os.path.exists(ref_so):
        import torch
        torch.set_num_threads(a.threads)
        torch.ops.load_library(ref_so)
        xt, got, wt = torch.from_numpy(x), torch.from_numpy(go), torch.from_numpy(w)
        b = torch.tensor([0, a.hw, 0, a.hw, 0, 1], dtype=torch.int32)
        fwd = torch.ops.torchshifts._shift2d_forward
        bwd = torch.ops.torchshifts._shift2d_backward
        times = []
        for it in range(a.iters + 1):
            t0 = time.perf_counter()
            out = fwd(xt, wt, b, list(shape), a.pad, False)
            t1 = time.perf_counter()
            gx, gw = bwd(got, wt, xt, b, a.pad, False)
            t2 = time.perf_counter()
            if it > 0:
                times.append((t1 - t0, t2 - t1))
        kind = "reference"
    else:
        from oracle import oracle as O
        times = []
        for it in range(a.iters):
            t0 = time.perf_counter()
            O.forward(x, w, a.pad, False)
            t1 = time.perf_counter()
            O.backward(go, w, x, a.pad, False)
            t2 = time.perf_counter()
            times.append((t1 - t0, t2 - t1))
        kind = "port"
        a.threads = 1
    f = min(t[0] for t in times)
    bk = min(t[1] for t in times)
    print(json.dumps({"value": elems / (f + bk) / 1e9, "unit": "Gelem/s", "cores": a.threads, "kind": kind,
                      "sample": "Shift2d SSL fwd+bwd N%d C%d %dx%d fp32 pad %d, best of %d (fwd %.1f ms, bwd %.1f ms)"
                                % (a.n, a.c, a.hw, a.hw, a.pad, len(times), f * 1e3, bk * 1e3)}))


if __name__ == "__main__":
    main()
