#!/usr/bin/env python3
"""Time the REAL reference CPU kernels (oracle/_ref/_C.so) -- bench.py's cpu_baseline leg.

TEST/BENCH INFRASTRUCTURE ONLY.  Run as a child process (it registers the same `torchshifts::*`
op names as the product library, so the two cannot share a process).  Prints one JSON line.
Exits 3 with "kind": "unavailable" when oracle/_ref is not present (--allow-port times the plain-C restatement
instead, labelled "port").
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


def host_cpu():
    model, logical = "unknown", os.cpu_count() or 1
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    return model, logical, usable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--c", type=int, default=256)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--pad", type=int, default=0)
    ap.add_argument("--both", action="store_true",
                    help="SURVEY 8d: time 1 thread AND all usable host cores on the --n sample, plus the N=4 slice")
    ap.add_argument("--full", type=int, default=0,
                    help="with --both: also time the FULL workload at this batch size (BASELINE config 2: 64), one timed "
                         "iteration after a warm-up pass, 1 thread and all cores (SURVEY 8d, BASELINE.md section 4)")
    ap.add_argument("--allow-port", action="store_true",
                    help="time the plain-C restatement (oracle/shift_oracle.c) when oracle/_ref is missing")
    a = ap.parse_args()
    ref_so = os.path.join(HERE, "_ref", "_C.so")
    model, logical, usable = host_cpu()
    if not os.path.exists(ref_so):
        if not a.allow_port:
            # never silently compare against a different CPU code (round-1 verdict)
            print(json.dumps({"value": None, "unit": "Gelem/s", "cores": 0, "kind": "unavailable",
                              "sample": "oracle/_ref/_C.so is missing (built by oracle/build_ref.sh where "
                                        "/root/reference exists); pass --allow-port to time the C restatement",
                              "host_cpu": model, "host_cores": logical}))
            sys.exit(3)
        from oracle import oracle as O

        def run(n, threads, iters):
            shape = (n, a.c, a.hw, a.hw)
            x, go = synth(shape, 1), synth(shape, 2)
            w = (synth((a.c, 2), 3) * 6 - 3).astype(np.float32)
            ts = []
            for _ in range(iters):
                t0 = time.perf_counter()
                O.forward(x, w, a.pad, False)
                t1 = time.perf_counter()
                O.backward(go, w, x, a.pad, False)
                ts.append((t1 - t0, time.perf_counter() - t1))
            return ts
        kind = "port"
        thread_sets = [1]
    else:
        import torch
        torch.ops.load_library(ref_so)
        fwd = torch.ops.torchshifts._shift2d_forward
        bwd = torch.ops.torchshifts._shift2d_backward

        def run(n, threads, iters):
            torch.set_num_threads(threads)
            shape = (n, a.c, a.hw, a.hw)
            if n > 8 and n % 8 == 0:  # the hash stream of 8 samples, repeated: same work, a fraction of the set-up time
                base = (8, a.c, a.hw, a.hw)
                xt = torch.from_numpy(synth(base, 1)).repeat(n // 8, 1, 1, 1)
                got = torch.from_numpy(synth(base, 2)).repeat(n // 8, 1, 1, 1)
            else:
                xt, got = torch.from_numpy(synth(shape, 1)), torch.from_numpy(synth(shape, 2))
            wt = torch.from_numpy((synth((a.c, 2), 3) * 6 - 3).astype(np.float32))
            b = torch.tensor([0, a.hw, 0, a.hw, 0, 1], dtype=torch.int32)
            ts = []
            for it in range(iters + 1):  # first pass = warm-up (first touch of fresh outputs)
                t0 = time.perf_counter()
                fwd(xt, wt, b, list(shape), a.pad, False)
                t1 = time.perf_counter()
                bwd(got, wt, xt, b, a.pad, False)
                t2 = time.perf_counter()
                if it > 0:
                    ts.append((t1 - t0, t2 - t1))
            return ts
        kind = "reference"
        thread_sets = [1, usable] if a.both else [a.threads]

    def entry(n, threads, iters):
        ts = run(n, threads, iters)
        f, bk = min(t[0] for t in ts), min(t[1] for t in ts)
        elems = n * a.c * a.hw * a.hw
        return {"value": elems / (f + bk) / 1e9, "unit": "Gelem/s", "cores": threads, "kind": kind,
                "sample": "Shift2d SSL fwd+bwd N%d C%d %dx%d fp32 pad %d, best of %d (fwd %.1f ms, bwd %.1f ms)"
                          % (n, a.c, a.hw, a.hw, a.pad, len(ts), f * 1e3, bk * 1e3)}

    runs = [entry(a.n, t, a.iters) for t in thread_sets]
    out = dict(runs[-1])  # the all-cores run when --both, else the requested thread count
    out["host_cpu"], out["host_cores"], out["usable_cores"] = model, logical, usable
    if a.both:
        out["single_thread"] = runs[0]
        if kind == "reference":
            out["note"] = ("all-cores run = the reference's at::parallel_for over N*C; its multi-thread weight gradient "
                           "races (global_scope.h:22) -- timing only")
        out["n4_slice"] = [entry(4, t, 1) for t in thread_sets]  # beside SURVEY section 6's N=4 numbers
        if a.full > 0 and kind == "reference":
            out["full_size"] = [entry(a.full, t, 1) for t in thread_sets]  # the stated configuration itself
    print(json.dumps(out))


if __name__ == "__main__":
    main()
