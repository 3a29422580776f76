#!/usr/bin/env python3
"""bench.py -- headline benchmark of the shiftnd hot path on MI355X.

Metric (BASELINE.json): Gelem/s + achieved HBM GB/s of Shift2d SSL fwd+bwd at N64 / C256 / 224x224
fp32 (BASELINE config 2), per GPU; with --gpus N the batch is sharded N ways (weak scaling: every
rank runs the full per-GPU config, no data-path collective -- the op is per-sample independent).

A "step" = one forward + one backward of the op over the synthetic batch, issued through the
drop-in dispatcher ops torch.ops.torchshifts._shift2d_forward/_backward (allocation of the outputs
and the workspace included).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline      the dominant kernel (plane_backward): algorithmic bytes (3*s per element, SURVEY
                section 8d) / average launch duration measured with HIP events on the launch
                stream, against the 8 TB/s HBM peak
  cpu_baseline  the REAL reference CPU kernels (oracle/_ref, built from /root/reference) timed on
                this host in a child process on a bounded sample (rank 0, --gpus 1 only)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--pad 0..4]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (nd, shape per GPU, dtype, active, description)
    "c2": (2, (64, 256, 224, 224), "float32", False, "Shift2d SSL fwd+bwd N64 C256 224x224 fp32"),
    "c3": (3, (8, 128, 16, 112, 112), "bfloat16", True, "Shift3d active fwd+bwd N8 C128 16x112x112 bf16"),
    "c4": (2, (128, 512, 56, 56), "quint8", False, "quantized Shift2d forward N128 C512 56x56 quint8"),
    "c5": (2, (64, 512, 224, 224), "float16", False, "Shift2d SSL fwd+bwd N64 C512 224x224 fp16 (per GPU)"),
}


def shard_range(n, rank, world):
    """Contiguous batch slice [lo, hi) of rank `rank`: the op is per-sample independent, so the N-GPU job
    is N independent slices of the batch (no data-path collective; sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def cpu_baseline():
    """must run before this process touches the GPU (child process; see oracle/ref_bench.py)"""
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), "--n", "8", "--threads", "1",
                              "--iters", "2"], capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "Gelem/s", "cores": 0, "kind": "unavailable", "sample": repr(e)[:200]}


def synth_tensor(torch, shape, seed, device, dtype, lo=0.0, hi=1.0):
    """repo-defined integer hash -> uniform [lo, hi): identical bytes on every box (no torch RNG)"""
    n = 1
    for s in shape:
        n *= s
    out = torch.empty(n, dtype=dtype, device=device)
    chunk = 1 << 26
    for start in range(0, n, chunk):
        m = min(chunk, n - start)
        idx = torch.arange(start, start + m, dtype=torch.int64, device=device)
        h = (idx * 2654435761 + seed) & 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x45D9F3B) & 0xFFFFFFFF
        h ^= h >> 16
        u = (h >> 8).to(torch.float32) / float(1 << 24)
        out[start:start + m] = (u * (hi - lo) + lo).to(dtype)
    return out.view(*shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--pad", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    base = None
    if rank == 0 and world == 1 and a.gpus == 1 and not a.no_cpu_baseline:
        base = cpu_baseline()  # before any GPU initialisation in this process

    import torch
    import torch.distributed as dist

    import torchshifts  # noqa: F401  (registers torch.ops.torchshifts.*; raises later if _C.so is missing)
    from torchshifts import abi
    from torchshifts.extension import _assert_has_ops
    _assert_has_ops()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the measured path)")
    # one process per GPU.  SHIFTND_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with
    # fewer GPUs than ranks (ranks then share devices; collectives run on host tensors) -- a functional check only.
    backend = os.environ.get("SHIFTND_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    nd, shape, dtname, active, desc = WORKLOADS[a.workload]
    # weak scaling: the global batch is world x the per-GPU batch; this rank owns one contiguous slice
    lo, hi = shard_range(shape[0] * world, rank, world)
    shape = (hi - lo,) + tuple(shape[1:])
    quant = dtname == "quint8"
    C = shape[1]
    elems = 1
    for s in shape:
        elems *= s
    ops = torch.ops.torchshifts
    fwd_op = getattr(ops, "_shift%dd_forward" % nd)
    bwd_op = getattr(ops, "_shift%dd_backward" % nd)
    borders = torch.tensor(abi.default_borders(torch.empty(shape, device="meta")), dtype=torch.int32)  # host

    # ---- synthetic inputs (resident before timing) ----------------------------------------------------
    seed = 1000 * rank
    w32 = synth_tensor(torch, (C, nd), seed + 3, dev, torch.float32, -3.0, 3.0)
    special = [0.0, 0.5, -1.5, 2.5, float(shape[2] + 3)]  # alignment classes + a beyond-the-dim shift
    for i, v in enumerate(special[:C]):
        w32[i, :] = v
    if quant:
        x = (synth_tensor(torch, shape, seed + 1, dev, torch.float32) * 255).to(torch.uint8)
        xq = torch._make_per_tensor_quantized_tensor(x, 1 / 255.0, 0)
        wq = torch.quantize_per_tensor(w32, 1.0, 128, torch.quint8)
        esize = 1
    else:
        dtype = getattr(torch, dtname)
        x = synth_tensor(torch, shape, seed + 1, dev, dtype)
        go = synth_tensor(torch, shape, seed + 2, dev, dtype)
        w = w32.to(dtype)
        esize = x.element_size()
    torch.cuda.synchronize()

    def step():
        if quant:
            return fwd_op(xq, wq, borders, list(shape), a.pad, False)
        out = fwd_op(x, w, borders, list(shape), a.pad, active)
        gx, gw = bwd_op(go, w, x, borders, a.pad, active)
        return out, gx, gw

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3

    # ---- per-kernel durations: HIP events on the launch stream, kernels called through the C ABI -------
    def event_time(fn, iters):
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters  # ms

    kiters = max(5, a.steps)
    kernels = {}
    if quant:
        xi = xq.int_repr()
        wi = wq.int_repr()
        outb = torch.empty_like(xi)
        t_f = event_time(lambda: abi.forward_quantized(xi, wi, 128, 0, a.pad, out=outb), kiters)
        qname = abi.last_kernel()
        kernels[qname] = {"ms": t_f, "GB/s": 2 * esize * elems / t_f / 1e6}
        dom_name, dom_ms, dom_bytes = qname, t_f, 2 * esize * elems
    else:
        outb, gxb, gwb = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
        ws = abi.backward_workspace(x, a.pad, active)
        t_f = event_time(lambda: abi.forward(x, w, a.pad, active, out=outb), kiters)
        t_b = event_time(lambda: abi.backward(go, w, x, a.pad, active, grad_x=gxb, grad_w=gwb, workspace=ws), kiters)
        abi.forward(x, w, a.pad, active, out=outb)
        fname = abi.last_kernel()
        abi.backward(go, w, x, a.pad, active, grad_x=gxb, grad_w=gwb, workspace=ws)
        bname = abi.last_kernel()
        kernels[fname] = {"ms": t_f, "GB/s": 2 * esize * elems / t_f / 1e6}
        kernels[bname] = {"ms": t_b, "GB/s": 3 * esize * elems / t_b / 1e6}
        dom_name, dom_ms, dom_bytes = bname, t_b, 3 * esize * elems
    path = "+".join(sorted(set(k.split("_")[0] for k in kernels)))

    if rank == 0:
        achieved = dom_bytes / dom_ms / 1e6  # GB/s
        # HBM bytes per launch of the dominant kernel from the PMC passes of this same command (profiles/collect.sh
        # cannot run inside the timed process: counters need their own rocprofv3 runs); null when no profile of this
        # workload is committed
        traffic, traffic_src = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_%s_traffic.json" % a.workload)
        if a.pad == 0 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                t = tj["per_launch"].get(dom_name)
                if t and "read_bytes" in t and "written_bytes" in t:
                    traffic, traffic_src = t["read_bytes"] + t["written_bytes"], "profiles/" + os.path.basename(tpath)
            except (OSError, ValueError, KeyError):
                pass
        step_bytes = (2 if quant else 5) * esize * elems
        result = {
            "metric": "Gelem/s, Shift2d fwd+bwd N64/C256/224x224" if a.workload == "c2" else "Gelem/s, " + desc,
            "value": elems * world / (ms_per_step * 1e-3) / 1e9,
            "unit": "Gelem/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"float32": "f32", "bfloat16": "bf16", "float16": "f16", "quint8": "u8"}[dtname],
            "data": "synthetic",
            "config": {"workload": desc + ", padding %d, per GPU; batch sharded over %d GPU(s), no collectives"
                                   % (a.pad, world),
                       "path": "torch.ops.torchshifts._shift%dd_forward/_backward -> libshiftnd_hip.so (%s kernels)"
                               % (nd, path)},
            "achieved_hbm_GBps_step": step_bytes / (ms_per_step * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_kernel_ms": dom_ms, "algorithmic_bytes": dom_bytes},
            "kernels": kernels,
        }
        if base is not None:
            result["cpu_baseline"] = base
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
