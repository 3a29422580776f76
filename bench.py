#!/usr/bin/env python3
"""bench.py -- headline benchmark of the shiftnd hot path on MI355X.

Metric (BASELINE.json): Gelem/s + achieved HBM GB/s of Shift2d SSL fwd+bwd at N64 / C256 / 224x224
fp32 (BASELINE config 2), per GPU; with --gpus N the batch is sharded N ways (weak scaling: every
rank runs the full per-GPU config, no data-path collective -- the op is per-sample independent).

A "step" = one forward + one backward of the op over the synthetic batch, issued through the
drop-in dispatcher ops torch.ops.torchshifts._shift2d_forward/_backward (allocation of the outputs
and the workspace included).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline      the dominant kernel (step_backward on C2): algorithmic bytes (3*s per element, SURVEY
                section 8d) / average launch duration measured with HIP events on the launch
                stream, against the 8 TB/s HBM peak
  cpu_baseline  the REAL reference CPU kernels (oracle/_ref, built from /root/reference) timed on
                this host in a child process on a bounded sample (rank 0, --gpus 1 only)

  roofline.box_stream / frac_of_box   the same box's plain 1R1W / 2R1W float4 streams (tools/stream_probe, a child
                process that has exited before this one touches the GPU) and the dominant kernel's rate against them

  configs       (default run only: --gpus 1, workload c2, padding 0) AFTER the headline's timed region, in the same
                process, every other BASELINE config (C2 paddings 1-4, C3 paddings 0-4, C4, C5 per GPU) and the round's
                other workloads are built, timed for >= 20 steps through the same dispatcher ops, measured kernel by
                kernel with HIP events and freed again.  --no-configs skips it.

OUTPUT (rank 0), in this order:
  1. `{"bench_detail": ...}`    one line: everything measured (per-kernel medians, the box calibration, every rank's device,
                                the whole cpu_baseline record, every config's record); also written to
                                gpurun_out/bench_detail.json when that directory exists
  2. the RESULT line, LAST and never longer than HEADLINE_MAX_BYTES (a consumer that keeps only a few KB of the
     tail of stdout must still find one whole line): the contract's fields + `roofline` + `cpu_baseline` (trimmed; with
     `own_cpu_key` = this repo's CPU dispatch key timed beside the reference) + `configs` = {name: [ms_per_step,
     dominant kernel, its fraction of 8 TB/s, counter traffic / algorithmic bytes or null]}

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--pad 0..4]      (NAME: see WORKLOADS)

Multi-GPU: `python bench.py --gpus N` starts the N rank processes itself (one per GPU; the parent never
touches the GPU); under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)
the ranks come from RANK / LOCAL_RANK / WORLD_SIZE.  BASELINE config 5 (N512 C512 224x224 fp16 over 8 GPUs)
is `python bench.py --workload c5 --gpus 8`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (the guide's measured float4 copy: 6.29 TB/s; tools/stream_probe
#                         measures 6.4-6.5 TB/s on this pool's boxes and the line carries that number as roofline.box_stream)

WORKLOADS = {
    # name: (nd, shape per GPU, dtype, active, description)
    "c2": (2, (64, 256, 224, 224), "float32", False, "Shift2d SSL fwd+bwd N64 C256 224x224 fp32"),
    "c3": (3, (8, 128, 16, 112, 112), "bfloat16", True, "Shift3d active fwd+bwd N8 C128 16x112x112 bf16"),
    "c4": (2, (128, 512, 56, 56), "quint8", False, "quantized Shift2d forward N128 C512 56x56 quint8"),
    "c5": (2, (64, 512, 224, 224), "float16", False, "Shift2d SSL fwd+bwd N64 C512 224x224 fp16 (per GPU)"),
    # not a BASELINE config: C2's tensor through the interpolating (active) kernels
    "c2a": (2, (64, 256, 224, 224), "float32", True, "Shift2d active fwd+bwd N64 C256 224x224 fp32"),
    # round 4 -- the reference's everyday cases: cropped windows (every emulate_dw module with padding < kernel // 2,
    # modules/shifts.py:41-46) and Shift1d (functional.py:7-36).  t1 / t1a = the reference's own test script
    # (tests/shifts_test.py:9-28: N512 C16 64x64, 3x3 depthwise emulation without padding -> cut 1 / 1, output 62x62)
    "c2crop": (2, (64, 256, 224, 224), "float32", False, "Shift2d SSL fwd+bwd N64 C256 224x224 fp32, cut [[1,1],[1,1]] (output 222x222)"),
    "c2acrop": (2, (64, 256, 224, 224), "float32", True, "Shift2d active fwd+bwd N64 C256 224x224 fp32, cut [[1,1],[1,1]] (output 222x222)"),
    "t1": (2, (512, 16, 64, 64), "float32", False, "Shift2d SSL fwd+bwd N512 C16 64x64 fp32, cut [[1,1],[1,1]] (the reference's tests/shifts_test.py)"),
    "t1a": (2, (512, 16, 64, 64), "float32", True, "Shift2d active fwd+bwd N512 C16 64x64 fp32, cut [[1,1],[1,1]] (the reference's tests/shifts_test.py)"),
    "c1d": (1, (256, 512, 4096), "float32", False, "Shift1d SSL fwd+bwd N256 C512 L4096 fp32"),
    "c1da": (1, (256, 512, 4096), "float32", True, "Shift1d active fwd+bwd N256 C512 L4096 fp32"),
    "c1dh": (1, (256, 512, 4096), "float16", False, "Shift1d SSL fwd+bwd N256 C512 L4096 fp16"),
    # round 5 -- rows that are not whole 16-byte pieces: the deep stages of an ImageNet CNN (14x14), and the OUTPUT of a cropped
    # shift as the next layer's input (62x62 = t1's output, 222x222 = c2crop's output; modules/shifts.py:41-46)
    "r14": (2, (128, 1024, 14, 14), "float32", False, "Shift2d SSL fwd+bwd N128 C1024 14x14 fp32 (ragged rows)"),
    "r14a": (2, (128, 1024, 14, 14), "float32", True, "Shift2d active fwd+bwd N128 C1024 14x14 fp32 (ragged rows)"),
    "r7": (2, (128, 2048, 7, 7), "float32", False, "Shift2d SSL fwd+bwd N128 C2048 7x7 fp32 (ragged rows)"),
    "r14h": (2, (128, 1024, 14, 14), "float16", False, "Shift2d SSL fwd+bwd N128 C1024 14x14 fp16 (ragged rows)"),
    "r62": (2, (512, 16, 62, 62), "float32", False, "Shift2d SSL fwd+bwd N512 C16 62x62 fp32 (ragged rows: t1's output as the next input)"),
    "r222": (2, (64, 256, 222, 222), "float32", False, "Shift2d SSL fwd+bwd N64 C256 222x222 fp32 (ragged rows: c2crop's output as the next input)"),
    "r225": (2, (8, 64, 225, 225), "float32", False, "Shift2d SSL fwd+bwd N8 C64 225x225 fp32 (ragged rows)"),
    # channels-last tensors as they lie: saved input, incoming gradient and grad_x all NHWC / NDHWC
    "cl2d": (2, (16, 256, 224, 224), "float32", False, "Shift2d SSL fwd+bwd N16 C256 224x224 fp32, channels-last (NHWC) tensors"),
    "cl2da": (2, (16, 256, 224, 224), "float32", True, "Shift2d active fwd+bwd N16 C256 224x224 fp32, channels-last (NHWC) tensors"),
    "cl3d": (3, (8, 128, 16, 112, 112), "float32", False, "Shift3d SSL fwd+bwd N8 C128 16x112x112 fp32, channels-last (NDHWC) input, public op + autograd"),
    "cl3da": (3, (8, 128, 16, 112, 112), "float32", True, "Shift3d active fwd+bwd N8 C128 16x112x112 fp32, channels-last (NDHWC) input, public op + autograd"),
    "cl3dh": (3, (8, 128, 16, 112, 112), "bfloat16", True, "Shift3d active fwd+bwd N8 C128 16x112x112 bf16, channels-last (NDHWC) input, public op + autograd"),
    # round 6 -- the module's tail (modules/shifts.py:81-89,150-153): emulate_dw = {kernel_size 3, stride 2,
    # padding 0} -> cut [[1,1],...] and avg_pool(kernel = stride = 2, ceil_mode) over the shifted window, as ONE pass
    # through torch.ops.torchshifts._shift{N}d_pool_forward/_backward
    "c2pool": (2, (64, 256, 224, 224), "float32", False, "Shift2d SSL + avg_pool 2 fwd+bwd N64 C256 224x224 fp32, cut [[1,1],[1,1]] (emulate_dw k3 s2 p0)"),
    "c3pool": (3, (8, 128, 16, 112, 112), "bfloat16", True, "Shift3d active + avg_pool 2 fwd+bwd N8 C128 16x112x112 bf16, cut 1/1 per dim (emulate_dw k3 s2 p0)"),
    "c4pool": (2, (128, 512, 56, 56), "quint8", False, "quantized Shift2d + avg_pool 2 forward N128 C512 56x56 quint8, cut [[1,1],[1,1]] (emulate_dw k3 s2 p0)"),
}

# (c3crop: BASELINE config 3's tensor behind emulate_dw {kernel 3, stride 1, padding 0} -- the cut without the pool; c3pool: with stride 2)
WORKLOADS["c3crop"] = (3, (8, 128, 16, 112, 112), "bfloat16", True, "Shift3d active fwd+bwd N8 C128 16x112x112 bf16, cut 1/1 per dim (output 14x110x110)")

# fused shift + average pool workloads: the pool's kernel = stride
POOLS = {"c2pool": 2, "c3pool": 2, "c4pool": 2}

# workloads whose tensors are channels-last (dense NHWC / NDHWC strides behind the logical NCHW / NCDHW shape)
CHANNELS_LAST = {"cl2d", "cl2da", "cl3d", "cl3da", "cl3dh"}

# What the default run times after the headline (same process, >= 20 steps each): every BASELINE.json config the headline is not
# -- C2 paddings 1-4, C3 paddings 0-4, C4, C5 (per GPU) -- then the round's other workloads.
EXTRA_CONFIGS = ([("c2_pad%d" % p, "c2", p) for p in (1, 2, 3, 4)] + [("c3_pad%d" % p, "c3", p) for p in range(5)] +
                 [("c4", "c4", 0), ("c5", "c5", 0)] +
                 [(n, n, 0) for n in ("c2a", "c2crop", "c2acrop", "t1", "t1a", "c1d", "r14", "r14a", "r62", "r222", "cl2d", "cl2da", "cl3d", "cl3da",
                                      "c2pool", "c3pool", "c4pool", "c3crop")])
BASELINE_CONFIGS = [c[0] for c in EXTRA_CONFIGS[:11]]  # what BASELINE.json's `configs` list beyond the headline
HEADLINE_MAX_BYTES = 4000  # the RESULT line is the last line of stdout and fits a small tail (round-5 verdict: a 24 KB line was lost)
ESIZE = {"float32": 4, "bfloat16": 2, "float16": 2, "quint8": 1}

# user `borders` of the cropped workloads ([nD, 2] cut-left / cut-right amounts, functional.py:22,32-35)
CUTS = {"c2crop": [[1, 1], [1, 1]], "c2acrop": [[1, 1], [1, 1]], "t1": [[1, 1], [1, 1]], "t1a": [[1, 1], [1, 1]],
        "c2pool": [[1, 1], [1, 1]], "c3pool": [[1, 1], [1, 1], [1, 1]], "c4pool": [[1, 1], [1, 1]], "c3crop": [[1, 1], [1, 1], [1, 1]]}


def shard_range(n, rank, world):
    """Contiguous batch slice [lo, hi) of rank `rank`: the op is per-sample independent, so the N-GPU job
    is N independent slices of the batch (no data-path collective; sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def cpu_baseline(workload_pad=0):
    """The REAL reference CPU kernels on this host (oracle/ref_bench.py, child process, before this process touches
    the GPU).  SURVEY section 8d: timed with 1 thread AND with all host cores, host CPU model and core count stated.
    `value`/`cores` = the all-cores run (the reference's at::parallel_for over N*C); `single_thread` beside it (the
    only run whose weight gradient is race-free in the reference, global_scope.h:22)."""
    script = os.path.join(ROOT, "oracle", "ref_bench.py")
    try:
        out = subprocess.run([sys.executable, script, "--n", "24", "--pad", str(workload_pad), "--iters", "2", "--both",
                              "--full", "64"], capture_output=True, text=True, timeout=900)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            raise RuntimeError("ref_bench.py exit %d: %s" % (out.returncode, (out.stderr or out.stdout)[-300:]))
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "Gelem/s", "cores": 0, "kind": "unavailable", "sample": repr(e)[:300]}


def own_cpu_key(pad=0, n=8):
    """This repo's own CPU dispatch key (csrc/torch_cpu_backend.cpp, all host cores) on a bounded slice of the headline workload,
    beside the reference's: `bench.py --device cpu` in a child process, before this one touches the GPU."""
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--device", "cpu", "--shape", "%d,256,224,224" % n, "--pad",
                              str(pad), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            raise RuntimeError("exit %d: %s" % (out.returncode, (out.stderr or out.stdout)[-200:]))
        j = json.loads(lines[-1])
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        return {"value": j["value"], "unit": "Gelem/s", "cores": cores, "kind": "port",
                "sample": "torch_cpu_backend.cpp, N%d C256 224x224 fp32 pad %d, mean of 2 steps (%.1f ms/step)" % (n, pad, j["ms_per_step"])}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "kind": "unavailable", "sample": repr(e)[:200]}


def _sig(v, digits=5):
    """floats to `digits` significant digits (the RESULT line is size-bounded), containers recursively"""
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def fallback_tail():
    """Share of a seeded sample of 6 000 geometries (tools/route_census.py: 1-3 dims, rows of 1..70 000 elements, windows, channels-last,
    every dtype; through the op's layout rule) that the one-thread-per-element fallback kernels served, from the newest committed
    census (profiles/r<NN>_route_census.txt: it needs its own GPU run).  None when no census carries the summary lines."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_route_census.txt")):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None
    out = {}
    try:
        for line in open(best[1]):
            m = re.match(r"fallback tail .*: (\w+)\s+(\d+) of\s+(\d+) =", line)
            if m:
                out[m.group(1)] = [int(m.group(2)), int(m.group(3))]
    except OSError:
        return None
    if not out:
        return None
    out["source"] = "profiles/" + os.path.basename(best[1])
    return out


def headline_line(result, limit=None):
    """The RESULT line from the full record: the contract's fields, `roofline`, a trimmed `cpu_baseline`, compact `configs`;
    never longer than HEADLINE_MAX_BYTES (optional parts are dropped in a fixed order until it fits)."""
    limit = limit or HEADLINE_MAX_BYTES
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_min",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    h = {k: result[k] for k in keep if k in result}
    cfg = result.get("config", {})
    ranks = dict(cfg.get("ranks", {}))
    per_rank = ranks.pop("per_rank", [])
    ranks["per_rank"] = [[r.get("rank"), r.get("host"), r.get("device_index"), r.get("pci_bus_id")] for r in per_rank]
    h["config"] = {"workload": cfg.get("workload"), "path": cfg.get("path"), "ranks": ranks}
    h["per_rank_ms"] = result.get("per_rank_ms")
    h["achieved_hbm_GBps_step"] = result.get("achieved_hbm_GBps_step")
    h["kernels"] = {k: {f: v[f] for f in ("ms", "GB/s", "frac_of_box") if f in v} for k, v in (result.get("kernels") or {}).items()}
    rl = result.get("roofline")
    if rl is not None:
        rl = dict(rl)
        box = rl.pop("box_stream", None) or {}
        rl.pop("traffic_source", None)
        if box.get("2R1W_GBps"):
            rl["box_GBps"] = {"1R1W": box.get("1R1W_GBps"), "2R1W": box.get("2R1W_GBps")}
        src = result.get("roofline", {}).get("traffic_source")
        if src:
            rl["traffic_source"] = src
    h["roofline"] = rl
    if result.get("fallback_tail"):
        h["fallback_tail"] = result["fallback_tail"]   # {entry point: [problems on strided_* / cl_* fallback kernels, problems], source}
    base = result.get("cpu_baseline")
    if base is not None:
        b = {k: base.get(k) for k in ("value", "unit", "cores", "kind", "host_cpu", "sample")}
        st = base.get("single_thread")
        if st:
            b["single_thread"] = {"value": st.get("value"), "cores": st.get("cores")}
        full = base.get("full_size")
        if full:
            b["full_size"] = [{"value": f.get("value"), "cores": f.get("cores")} for f in full]
        own = base.get("own_cpu_key")
        if own:
            b["own_cpu_key"] = {"value": own.get("value"), "cores": own.get("cores"), "kind": own.get("kind")}
        h["cpu_baseline"] = b
    cfgs = result.get("configs")
    if cfgs is not None:
        comp = {}
        for name, c in cfgs.items():
            if "roofline" not in c:
                comp[name] = [None, "skipped" if "skipped" in c else "error", None, None]
                continue
            r = c["roofline"]
            ratio = (r["traffic"] / r["algorithmic_bytes"]) if r.get("traffic") and r.get("algorithmic_bytes") else None
            comp[name] = [float("%.4g" % c["ms_per_step"]), r["kernel"], float("%.3g" % r["frac"]),
                          None if ratio is None else float("%.3g" % ratio)]
        h["configs"] = comp
        h["configs_fields"] = "ms_per_step, dominant kernel, frac of 8 TB/s, counter traffic / algorithmic bytes"
        h["configs_wall_s"] = result.get("configs_wall_s")
    h = _sig(h)

    def dump():
        return json.dumps(h, separators=(",", ":"))
    # optional parts, least important first
    drops = [lambda: h.pop("configs_fields", None),
             lambda: h.pop("fallback_tail", None),
             lambda: h.__setitem__("configs", {k: v for k, v in h["configs"].items() if k in BASELINE_CONFIGS}) if "configs" in h else None,
             lambda: h["config"].pop("path", None),
             lambda: [v.pop("frac_of_box", None) for v in h["kernels"].values()],
             lambda: h["config"]["ranks"].pop("per_rank", None),
             lambda: (h.get("cpu_baseline") or {}).pop("full_size", None),
             lambda: (h.get("roofline") or {}).pop("traffic_source", None),
             lambda: h.pop("kernels", None),
             lambda: (h.get("cpu_baseline") or {}).pop("sample", None),
             lambda: h.pop("configs", None)]
    line = dump()
    for d in drops:
        if len(line) <= limit:
            break
        d()
        line = dump()
    assert len(line) <= limit, "RESULT line is %d bytes" % len(line)
    return line


def emit(result, out=None):
    """rank 0's output: the detail line first, the size-bounded RESULT line LAST."""
    out = out or sys.stdout
    detail = json.dumps({"bench_detail": result})
    out.write(detail + "\n")
    ddir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(ddir) and os.access(ddir, os.W_OK):
        try:
            with open(os.path.join(ddir, "bench_detail.json"), "w") as f:
                f.write(detail + "\n")
        except OSError:
            pass
    out.write(headline_line(result) + "\n")
    out.flush()


def flush_c_stdio():
    """RCCL announces itself with a C-level printf ("Librccl path : ...") that sits in the C library's buffer -- stdout is a pipe --
    until the process exits, i.e. AFTER the RESULT line Python has already written: the last line of stdout was RCCL's, not the
    result (found by tests/test_bench_ranks.py on the GPU box).  Every rank empties the C buffers right after the rendezvous and
    again before rank 0 prints."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def box_stream(tensor_bytes=None):
    """Same-box calibration: tools/stream_probe (a plain HIP binary, built by __graft_entry__.build()) streams C2-sized
    buffers with float4 accesses -- best 1-read-1-write and 2-read-1-write rates over a small fixed set of launch shapes --
    in a child process that has exited before this process touches the GPU.  The kernels' rates are reported against
    these next to the 8 TB/s fraction: boxes of the pool differ by several per cent, the ratio to the box's own plain
    streams does not."""
    exe = os.path.join(ROOT, "tools", "stream_probe")
    try:
        out = subprocess.run([exe] + (["--bytes", str(int(tensor_bytes))] if tensor_bytes else []), capture_output=True, text=True,
                             timeout=300)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            raise RuntimeError("stream_probe exit %d: %s" % (out.returncode, (out.stderr or out.stdout)[-300:]))
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        return {"probe": "unavailable", "error": repr(e)[:300]}


def newest_traffic(workload, kernel_name, pad):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of this workload
    (profiles/r<NN>_<workload>_traffic.json, written by profiles/collect.sh + summarize.py: counters need their own
    rocprofv3 passes, they cannot be read inside the timed process).  None when no profile names this kernel."""
    import glob
    import re
    best = None
    tags = ["%s_pad%d" % (workload, pad)] + ([workload] if pad == 0 else [])
    for f in [g for t in tags for g in glob.glob(os.path.join(ROOT, "profiles", "r*_%s_traffic.json" % t))]:
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None, None
    try:
        per = json.load(open(best[1]))["per_launch"]
        # (shiftnd_last_kernel names the ROUTE -- crop_backward_ragged, flat_gather_forward -- the profile the kernel function)
        t = None
        plain = kernel_name.replace("(pool)", "")
        for cand in (plain, plain.replace("_ragged", ""), plain.replace("_gather_", "_").replace("_active_", "_"),
                     plain.replace("_ncdhw_grad", "").replace("_nchw_grad", ""), plain.replace("_pool", ""),
                     plain.replace("_gather_", "_").replace("_active_", "_").replace("_rows", "3"), plain.replace("_sparse", ""), plain.replace("_crop", "").replace("_sparse", ""),
                     plain.replace("_crop", "").replace("_sparse", "").replace("_pool", "")):
            if cand in per:
                t = per[cand]
                break
        if t and "read_bytes" in t and "written_bytes" in t:
            return t["read_bytes"] + t["written_bytes"], "profiles/%s (round %d)" % (os.path.basename(best[1]), best[0])
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) and wait.  This parent
    never imports torch or touches the GPU; the children are fresh interpreters (no exec from a GPU process)."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(a.gpus)
    procs = []
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
    # supervise: the first rank that fails takes the others down (they would otherwise sit in the rendezvous or in a
    # barrier until the collective timeout), and the parent exits with its code within seconds
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                for q in live:
                    q.terminate()
                deadline = time.time() + 5.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
                break
    if rc:
        sys.stderr.write("bench.py: a rank exited with code %d; the other ranks were stopped\n" % rc)
    return rc


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


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--pad", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shape", default=None,
                    help="comma-separated per-GPU shape replacing the workload's (functional checks of the rank code "
                         "only: the JSON line then says so and is not a measurement of the named workload)")
    ap.add_argument("--allow-oversubscribe", action="store_true",
                    help="let --gpus N run on a box with fewer than N GPUs (ranks share devices over gloo): a functional "
                         "check of the rank code, never a scaling point.  Without it such a request is an error.")
    ap.add_argument("--no-probe", action="store_true", help="skip the same-box stream calibration (tools/stream_probe)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default run only: skip the other BASELINE configs / workloads timed after the headline")
    ap.add_argument("--configs", default=None, help="comma-separated subset of the configs timed after the headline")
    ap.add_argument("--configs-budget-s", type=float, default=150.0,
                    help="wall-clock budget of the configs leg; configs that would start beyond it are reported as skipped")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise the process group (RCCL on a GPU box) even for --gpus 1: exercises the rendezvous, "
                         "barrier, all_gather and all_reduce(MAX) calls of the multi-GPU path on a 1-GPU box (tests)")
    ap.add_argument("--device", default="cuda", choices=("cuda", "cpu"),
                    help="cpu = run the rank/launcher plumbing on the CPU dispatch key with gloo (tests only; "
                         "no roofline, never a result)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a, argv))  # parent: no torch, no GPU call

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with `python bench.py --gpus N`, or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)" % (a.gpus, world))

    default_run = rank == 0 and world == 1 and a.workload == "c2" and a.pad == 0 and a.shape is None and a.device == "cuda"
    base = None
    if rank == 0 and world == 1 and a.workload == "c2" and a.shape is None and a.device == "cuda" \
            and not a.no_cpu_baseline:
        base = cpu_baseline(a.pad)  # child process, before any GPU initialisation in this process
        base["own_cpu_key"] = own_cpu_key(a.pad)
    probes = {}   # same-box streams by size class: tensors within a quarter octave of each other share one calibration run

    def size_class(nbytes):
        import math
        return int(round(4 * math.log2(max(nbytes, 1))))

    def tensor_bytes(name, shape_arg=None):
        _nd, _shape, _dt, _act, _desc = WORKLOADS[name]
        _n = 1
        for _s in (tuple(int(v) for v in shape_arg.split(",")) if shape_arg else _shape):
            _n *= _s
        return _n * ESIZE[_dt]

    if rank == 0 and world == 1 and a.device == "cuda" and not a.no_probe:
        # child process too: it has left the GPU before this process initialises it.  Streams of THIS workload's tensor size
        # (a 0.2 ms kernel over 0.4 GB tensors pays launch ramp and tail that a 1.5 ms kernel over 3.3 GB does not); the
        # default run also calibrates at the tensor sizes of the BASELINE configs it times after the headline
        sizes = [tensor_bytes(a.workload, a.shape)]
        if default_run and not a.no_configs:
            sizes += [tensor_bytes(w) for _n, w, _p in EXTRA_CONFIGS]
        for nb in sizes:
            if size_class(nb) not in probes:
                probes[size_class(nb)] = box_stream(nb)
    probe = probes.get(size_class(tensor_bytes(a.workload, a.shape)))

    import torch
    import torch.distributed as dist

    import torchshifts  # noqa: F401  (registers torch.ops.torchshifts.*; raises later if _C.so is missing)
    from torchshifts import abi
    from torchshifts.extension import _assert_has_ops
    _assert_has_ops()

    on_gpu = a.device == "cuda"
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the measured path)")
    # One process per GPU over RCCL.  When the box has fewer GPUs than ranks (a 1-GPU box running `--gpus 2` as a
    # functional check of the rank code) the ranks share devices and rendezvous over gloo; the line says so.
    ndev = torch.cuda.device_count() if on_gpu else 0
    backend = os.environ.get("SHIFTND_BENCH_BACKEND") or ("nccl" if on_gpu and ndev >= world else "gloo")
    oversubscribed = on_gpu and ndev < world
    if oversubscribed and not a.allow_oversubscribe:
        raise SystemExit("bench.py: --gpus %d but this box has %d GPU(s); a scaling point needs one GPU per rank "
                         "(--allow-oversubscribe runs the ranks on shared devices as a functional check)" % (world, ndev))
    if on_gpu:
        dev_index = local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    else:
        dev = torch.device("cpu")
    use_pg = world > 1 or a.force_process_group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("SHIFTND_BENCH_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)
        flush_c_stdio()
    if os.environ.get("SHIFTND_BENCH_FAIL_RANK") == str(rank):  # tests: a rank that dies after the rendezvous
        sys.stderr.write("bench.py: rank %d exits on request (SHIFTND_BENCH_FAIL_RANK)\n" % rank)
        os._exit(7)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if use_pg:
            dist.barrier()
        sync()

    ops = torch.ops.torchshifts

    class Workload:
        """The synthetic tensors of one workload (resident in HBM), its step through the dispatcher ops, and its kernels
        through the C ABI."""

        def __init__(self, name, pad, shape_arg=None):
            self.name, self.pad = name, pad
            nd, shape, dtname, active, desc = WORKLOADS[name]
            if shape_arg:
                shape = tuple(int(v) for v in shape_arg.split(","))
                assert len(shape) == nd + 2, "--shape needs %d dims for workload %s" % (nd + 2, name)
                desc = "FUNCTIONAL CHECK shape %s of: %s" % (list(shape), desc)
            # weak scaling: the global batch is world x the per-GPU batch; this rank owns one contiguous slice
            lo, hi = shard_range(shape[0] * world, rank, world)
            shape = (hi - lo,) + tuple(shape[1:])
            self.nd, self.shape, self.dtname, self.active, self.desc = nd, shape, dtname, active, desc
            self.quant = quant = dtname == "quint8"
            C = shape[1]
            self.elems = 1
            for s_ in shape:
                self.elems *= s_
            self.fwd_op = getattr(ops, "_shift%dd_forward" % nd)
            self.bwd_op = getattr(ops, "_shift%dd_backward" % nd)
            self.cuts = cuts = CUTS.get(name)
            if cuts is None:
                self.bl, self.oshape = abi.default_borders(torch.empty(shape, device="meta")), list(shape)
            else:  # check_borders (ops/shifts.cpp:93-135): cut amounts -> the absolute [l, r) window and the output size
                self.bl, self.oshape = abi.check_borders(list(shape), cuts, nd)
            self.borders = torch.tensor(self.bl, dtype=torch.int32)  # host
            self.oelems = 1
            for s_ in self.oshape:
                self.oelems *= s_
            # fused shift + average pool (kernel = stride = pool, ceil_mode): the op's output / incoming gradient is the pooled window
            self.pool = POOLS.get(name)
            self.gshape = list(self.oshape)
            if self.pool:
                self.pools = [self.pool] * nd
                self.fwd_op = getattr(ops, "_shift%dd_pool_forward" % nd)
                self.bwd_op = getattr(ops, "_shift%dd_pool_backward" % nd)
                self.gshape = list(self.oshape[:2]) + [-(-v // self.pool) for v in self.oshape[2:]]
            self.pelems = 1
            for s_ in self.gshape:
                self.pelems *= s_
            # ---- synthetic inputs (resident before timing) ------------------------------------------------
            seed = 1000 * rank
            w32 = synth_tensor(torch, (C, nd), seed + 3, dev, torch.float32, -3.0, 3.0)
            special = [0.0, 0.5, -1.5, 2.5, float(shape[2] + 3)]  # alignment classes + a beyond-the-dim shift
            for i, v in enumerate(special[:C]):
                w32[i, :] = v
            cl = name in CHANNELS_LAST
            fmt = (torch.channels_last if nd == 2 else torch.channels_last_3d) if cl else torch.contiguous_format
            if quant:
                x = (synth_tensor(torch, shape, seed + 1, dev, torch.float32) * 255).to(torch.uint8)
                self.xq = torch._make_per_tensor_quantized_tensor(x, 1 / 255.0, 0)
                self.wq = torch.quantize_per_tensor(w32, 1.0, 128, torch.quint8)
                self.esize = 1
            else:
                dtype = getattr(torch, dtname)
                if not on_gpu and dtype in (torch.float16, torch.bfloat16):
                    dtype = torch.float32  # the CPU key serves float/double like the reference's (shifts_cpu.cpp:228)
                self.x = synth_tensor(torch, shape, seed + 1, dev, dtype).contiguous(memory_format=fmt)
                self.go = synth_tensor(torch, tuple(self.gshape), seed + 2, dev, dtype).contiguous(memory_format=fmt)
                self.w = w32.to(dtype)
                self.esize = self.x.element_size()
            self.fmt = fmt
            # NDHWC (round 6): the step a model takes -- the PUBLIC op on the channels_last_3d input, autograd for the backward, the
            # incoming gradient in the layout of the forward's output (NCDHW-contiguous, as the reference's float forward returns it,
            # cpu/shifts_cpu.cpp:221).  The autograd node changes the layout once and keeps the contiguous copy for its backward
            # (torch_binding.cpp: ShiftFunction<3>::forward); the private ops called one by one -- rounds 4-5's step -- change it twice
            # more (tools/ndhwc_autograd_bench.py: fp32 1.61 -> 1.01 ms).
            self.autograd = cl and nd == 3 and not quant and on_gpu
            if self.autograd:
                self.go = self.go.contiguous()
                self.xr, self.wr = self.x.detach().requires_grad_(True), self.w.detach().requires_grad_(True)
                self.pub_op = getattr(ops, "shift%dd" % nd)
                self.no_borders = torch.Tensor()
            sync()

        def step(self):
            if self.autograd:
                out = self.pub_op(self.xr, self.wr, self.no_borders, self.pad, self.active)
                gx, gw = torch.autograd.grad(out, [self.xr, self.wr], self.go)
                return out, gx, gw
            if self.pool:
                if self.quant:
                    return self.fwd_op(self.xq, self.wq, self.borders, self.oshape, self.pools, self.pad, False)
                out = self.fwd_op(self.x, self.w, self.borders, self.oshape, self.pools, self.pad, self.active)
                gx, gw = self.bwd_op(self.go, self.w, self.x, self.borders, self.pools, self.pad, self.active)
                return out, gx, gw
            if self.quant:
                return self.fwd_op(self.xq, self.wq, self.borders, self.oshape, self.pad, False)
            out = self.fwd_op(self.x, self.w, self.borders, self.oshape, self.pad, self.active)
            gx, gw = self.bwd_op(self.go, self.w, self.x, self.borders, self.pad, self.active)
            return out, gx, gw

        def step_bytes(self):
            """algorithmic bytes of one step.  Plain: forward x + out, backward grad_out + x + grad_x.  Pooled:
            forward reads x and writes the pooled window, backward reads the pooled gradient and x and writes grad_x."""
            if self.pool:
                fwd = self.esize * (self.elems + self.pelems)
                return fwd if self.quant else fwd + self.esize * (2 * self.elems + self.pelems)
            return 2 * self.esize * self.elems if self.quant else self.esize * (3 * self.elems + 2 * self.oelems)

        def kernel_times(self, kiters, probe):
            """per-kernel durations: HIP events on the launch stream, kernels called through the C ABI.
            Returns (kernels, dominant kernel's (name, ms, algorithmic bytes, stream kind))"""
            kernels = {}

            def record(name, t, nbytes, stream_kind=None):
                kernels[name] = {"ms": t[0], "median_ms": t[1], "min_ms": t[2], "GB/s": nbytes / t[0] / 1e6}
                ref = (probe or {}).get(stream_kind + "_GBps") if stream_kind else None
                if ref:  # against the plain stream of the same read / write mix on this box
                    kernels[name]["stream"] = stream_kind
                    kernels[name]["frac_of_box"] = nbytes / t[0] / 1e6 / ref
            esize, elems, oelems, pad, active = self.esize, self.elems, self.oelems, self.pad, self.active
            if self.pool:
                pelems, pools = self.pelems, self.pools
                if self.quant:
                    xi, wi = self.xq.int_repr(), self.wq.int_repr()
                    outb = torch.empty(self.gshape, dtype=xi.dtype, device=xi.device)
                    t_f = event_time(lambda: abi.forward_quantized_pooled(xi, wi, 128, 0, pad, pools, borders=self.bl, out=outb), kiters)
                    qname = abi.last_kernel()
                    record(qname, t_f, esize * (elems + pelems), "1R1W")
                    return kernels, (qname, t_f[0], esize * (elems + pelems), "1R1W")
                x, go, w = self.x, self.go, self.w
                outb, gxb, gwb = torch.empty(self.gshape, dtype=x.dtype, device=x.device), torch.empty_like(x), torch.empty_like(w)
                fbytes, bbytes = esize * (elems + pelems), esize * (2 * elems + pelems)
                try:
                    t_f = event_time(lambda: abi.forward_pooled(x, w, pad, active, pools, borders=self.bl, out=outb), kiters)
                    fname = abi.last_kernel() + "(pool)"
                except RuntimeError:   # geometry not served by the fused kernels: the op composes shift + avg_pool (timed as the op)
                    t_f = event_time(lambda: self.fwd_op(x, w, self.borders, self.oshape, pools, pad, active), kiters)
                    fname = "unfused:shift+avg_pool"
                try:
                    pws = torch.empty(abi.backward_pooled_workspace_bytes(x, pad, active, pools, self.bl), dtype=torch.uint8, device=x.device)
                    abi.backward_pooled(go, w, x, pad, active, pools, borders=self.bl, grad_x=gxb, grad_w=gwb, workspace=pws)
                    bname = abi.last_kernel() + "(pool)"
                    t_b = event_time(lambda: abi.backward_pooled(go, w, x, pad, active, pools, borders=self.bl, grad_x=gxb, grad_w=gwb,
                                                                 workspace=pws), kiters)
                except RuntimeError:
                    t_b = event_time(lambda: self.bwd_op(go, w, x, self.borders, pools, pad, active), kiters)
                    bname = "unfused:avg_pool_backward+shift"
                record(fname, t_f, fbytes, "1R1W")
                record(bname, t_b, bbytes, "2R1W")
                return kernels, (bname, t_b[0], bbytes, "2R1W")
            if self.quant:
                xi = self.xq.int_repr()
                wi = self.wq.int_repr()
                outb = torch.empty_like(xi)
                t_f = event_time(lambda: abi.forward_quantized(xi, wi, 128, 0, pad, out=outb), kiters)
                qname = abi.last_kernel()
                record(qname, t_f, 2 * esize * elems, "1R1W")
                return kernels, (qname, t_f[0], 2 * esize * elems, "1R1W")
            x, go, w = self.x, self.go, self.w
            if self.autograd:   # the kernels of the step above: one layout change, then the contiguous forward and backward
                xc, outb, gxb, gwb = torch.empty(x.shape, dtype=x.dtype, device=x.device), torch.empty_like(go), torch.empty_like(go), torch.empty_like(w)
                planes = x[0, 0].numel()

                def change():
                    abi.check(abi.lib().shiftnd_transpose(x.data_ptr(), xc.data_ptr(), x.shape[0], planes, x.shape[1], esize, abi._stream()),
                              "shiftnd_transpose")
                ws = abi.backward_workspace(xc, pad, active, None)
                t_t = event_time(change, kiters)
                t_f = event_time(lambda: abi.forward(xc, w, pad, active, out=outb), kiters)
                fname = abi.last_kernel()
                t_b = event_time(lambda: abi.backward(go, w, xc, pad, active, grad_x=gxb, grad_w=gwb, workspace=ws), kiters)
                bname = abi.last_kernel()
                record("transpose_tiles", t_t, 2 * esize * elems, "1R1W")
                record(fname, t_f, 2 * esize * elems, "1R1W")
                record(bname, t_b, 3 * esize * elems, "2R1W")
                return kernels, (bname, t_b[0], 3 * esize * elems, "2R1W")
            # (the float forward's output is NCHW-contiguous even for a channels-last input, cpu/shifts_cpu.cpp:221; grad_x
            # has the input's layout, :246)
            outb, gxb, gwb = torch.empty(self.oshape, dtype=x.dtype, device=x.device), torch.empty_like(x), torch.empty_like(w)
            bk = None if self.cuts is None else self.bl
            ws = abi.backward_workspace(x, pad, active, bk)
            t_f = event_time(lambda: abi.forward(x, w, pad, active, borders=bk, out=outb), kiters)
            t_b = event_time(lambda: abi.backward(go, w, x, pad, active, borders=bk, grad_x=gxb, grad_w=gwb, workspace=ws), kiters)
            abi.forward(x, w, pad, active, borders=bk, out=outb)
            fname = abi.last_kernel()
            abi.backward(go, w, x, pad, active, borders=bk, grad_x=gxb, grad_w=gwb, workspace=ws)
            bname = abi.last_kernel()
            # algorithmic bytes: forward reads x, writes out; backward reads grad_out and x, writes grad_x
            # (a cropped window: out / grad_out have the window's size)
            record(fname, t_f, esize * (elems + oelems), "1R1W")
            record(bname, t_b, esize * (2 * elems + oelems), "2R1W")
            return kernels, (bname, t_b[0], esize * (2 * elems + oelems), "2R1W")

    def event_time(fn, iters):
        stream = torch.cuda.current_stream()
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record(stream)
        for i in range(iters):
            fn()
            ev[i + 1].record(stream)
        ev[-1].synchronize()
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
        return ev[0].elapsed_time(ev[-1]) / iters, per[len(per) // 2], per[0]  # mean, median, min (ms)

    wl = Workload(a.workload, a.pad, a.shape)
    nd, shape, dtname, active, desc = wl.nd, wl.shape, wl.dtname, wl.active, wl.desc
    elems, quant = wl.elems, wl.quant
    step = wl.step

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = [dt / a.steps * 1e3]
    if use_pg:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(v.item()) / a.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3

    # ---- what every rank ran on (N > 1: all-gathered; the first 8-GPU run describes itself) ------------------------------
    def rank_info():
        info = {"rank": rank, "local_rank": local_rank, "host": os.uname().nodename, "pid": os.getpid(),
                "device_index": dev.index if on_gpu else None, "world_size": world, "backend": backend if use_pg else None}
        if on_gpu:
            pr = torch.cuda.get_device_properties(dev)
            info["device_name"] = pr.name
            info["uuid"] = str(getattr(pr, "uuid", "")) or None
            have_pci = all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
            info["pci_bus_id"] = ("%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)) if have_pci else None
            info["hbm_bytes"] = pr.total_memory
        if use_pg:
            info["pg_world_size"] = dist.get_world_size()
            info["pg_backend"] = dist.get_backend()
        return info

    per_rank = [rank_info()]
    if use_pg:
        every_info = [None] * world
        dist.all_gather_object(every_info, per_rank[0])
        per_rank = every_info
        if on_gpu and not oversubscribed:
            # one GPU per rank: abort only when two ranks PROVABLY share a device -- the same host and the same device index,
            # or the same meaningful uuid / PCI id (a torch build that reports an empty or all-zero uuid, or no PCI
            # attributes, says nothing: those fall back to the device index the rank selected)
            shared = distinct_device_conflicts(per_rank)
            if shared:
                raise SystemExit("bench.py: ranks share a GPU: %s -- %s" % (shared, per_rank))

    # ---- per-step spread (median / min over single steps; outside the contract's timed region) -----------
    singles = []
    for _ in range(min(a.steps, 50)):
        sync()
        t1 = time.perf_counter()
        step()
        sync()
        singles.append((time.perf_counter() - t1) * 1e3)
    singles.sort()

    kiters = max(5, a.steps)
    kernels = {}
    dom_name = dom_ms = dom_bytes = None
    if on_gpu:
        kernels, (dom_name, dom_ms, dom_bytes, dom_kind) = wl.kernel_times(kiters, probe)
    path = "+".join(sorted(set(k.split("_")[0] for k in kernels))) or "cpu key"

    def roofline_of(name, pad, dom, probe_, shape_arg=None):
        dname, dms, dbytes, dkind = dom
        achieved = dbytes / dms / 1e6  # GB/s
        traffic, traffic_src = newest_traffic(name, dname, pad) if not shape_arg else (None, None)
        r = {"bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
             "traffic_source": traffic_src, "avg_kernel_ms": dms,
             "algorithmic_bytes": dbytes}
        if probe_ is not None:
            ref = probe_.get(dkind + "_GBps")
            r["frac_of_box"] = achieved / ref if ref else None
        return r

    dtag = {"float32": "f32", "bfloat16": "bf16", "float16": "f16", "quint8": "u8"}
    if rank == 0:
        step_bytes = wl.step_bytes()
        result = {
            "metric": "Gelem/s, Shift2d fwd+bwd N64/C256/224x224" if a.workload == "c2" and not a.shape
                      else "Gelem/s, " + desc,
            "value": elems * world / (ms_per_step * 1e-3) / 1e9,
            "unit": "Gelem/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_median": singles[len(singles) // 2],
            "ms_per_step_min": singles[0],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtag[dtname],
            "data": "synthetic",
            "config": {"workload": desc + ", padding %d, per GPU; batch sharded over %d GPU(s), no collectives"
                                   % (a.pad, world),
                       "path": ("torch.ops.torchshifts._shift%dd_forward/_backward -> libshiftnd_hip.so (%s kernels)"
                                % (nd, path)) if on_gpu else "torch.ops.torchshifts (CPU dispatch key of _C.so)",
                       "ranks": {"world": world, "backend": backend if use_pg else None, "devices": ndev,
                                 "oversubscribed": oversubscribed,
                                 "device_name": torch.cuda.get_device_name(dev) if on_gpu else None,
                                 "per_rank": per_rank}},
            "per_rank_ms": per_rank_ms,
            "achieved_hbm_GBps_step": step_bytes / (ms_per_step * 1e-3) / 1e9,
            "kernels": kernels,
        }
        if dom_name is not None:
            result["roofline"] = roofline_of(a.workload, a.pad, (dom_name, dom_ms, dom_bytes, dom_kind), probe, a.shape)
            if probe is not None:
                fob = result["roofline"].pop("frac_of_box")
                result["roofline"]["box_stream"] = probe
                result["roofline"]["frac_of_box"] = fob
        else:
            result["roofline"] = None
        if not on_gpu:
            result["data"] = "synthetic; CPU functional check of the rank code, not a measurement"
        if oversubscribed:
            result["data"] = "synthetic; %d ranks share %d GPU(s): functional check, not a scaling point" % (world, ndev)
        if base is not None:
            result["cpu_baseline"] = base
        if default_run:
            result["fallback_tail"] = fallback_tail()

        # ---- the other BASELINE configs and workloads, driver-timed: same process, after the headline (whose tensors are
        # freed first), each built, warmed up, timed over >= 20 steps exactly like the headline, measured kernel by kernel
        if default_run and not a.no_configs:
            del wl, step
            configs = {}
            t_all = time.perf_counter()
            csteps, cwarm = max(20, a.steps), max(5, a.warmup)
            only = set(a.configs.split(",")) if a.configs else None
            for cname, wname, pad in EXTRA_CONFIGS:
                if only is not None and cname not in only and wname not in only:
                    continue
                if time.perf_counter() - t_all > a.configs_budget_s:
                    configs[cname] = {"skipped": "time budget (--configs-budget-s %g)" % a.configs_budget_s}
                    continue
                try:
                    torch.cuda.empty_cache()
                    cw = Workload(wname, pad)
                    for _ in range(cwarm):
                        cw.step()
                    barrier()
                    c0 = time.perf_counter()
                    for _ in range(csteps):
                        cw.step()
                    barrier()
                    cms = (time.perf_counter() - c0) / csteps * 1e3
                    cprobe = probes.get(size_class(tensor_bytes(wname)))
                    ck, cdom = cw.kernel_times(csteps, cprobe)
                    rl = roofline_of(wname, pad, cdom, cprobe)
                    configs[cname] = {
                        "workload": cw.desc + ", padding %d" % pad, "steps": csteps, "warmup": cwarm,
                        "ms_per_step": cms, "value": cw.elems / (cms * 1e-3) / 1e9, "unit": "Gelem/s", "dtype": dtag[cw.dtname],
                        "achieved_hbm_GBps_step": cw.step_bytes() / (cms * 1e-3) / 1e9,
                        "kernels": {k: {"ms": v["ms"], "GB/s": v["GB/s"], "frac": v["GB/s"] / HBM_PEAK_GBS,
                                        "frac_of_box": v.get("frac_of_box")} for k, v in ck.items()},
                        "roofline": {"kernel": rl["kernel"], "frac": rl["frac"], "frac_of_box": rl.get("frac_of_box"),
                                     "traffic": rl["traffic"], "traffic_source": rl["traffic_source"],
                                     "avg_kernel_ms": rl["avg_kernel_ms"], "algorithmic_bytes": rl["algorithmic_bytes"]}}
                    del cw, ck
                except Exception as e:  # noqa: BLE001  (one failing workload must not lose the headline line; it is reported)
                    configs[cname] = {"error": repr(e)[:300]}
            result["configs"] = configs
            result["configs_wall_s"] = time.perf_counter() - t_all
    # the RESULT line is the LAST thing any rank writes: the other ranks have left the process group (and flushed whatever the
    # C libraries buffered) before rank 0 prints
    if use_pg:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        emit(result)


def distinct_device_conflicts(per_rank):
    """Pairs of ranks that provably run on ONE GPU: the same host and the same selected device index, or the same
    meaningful uuid / PCI id.  An empty, all-zero or absent uuid and an absent PCI id say nothing (a ROCm / torch build may
    not report them): such ranks are compared by (host, device index) only."""
    def meaningful(v):
        return bool(v) and any(ch not in "0-:. " for ch in str(v).lower())
    seen, shared = {}, []
    for r in per_rank:
        keys = [("index", r["host"], r["device_index"])]
        if meaningful(r.get("uuid")):
            keys.append(("uuid", r["uuid"]))
        if meaningful(r.get("pci_bus_id")):
            keys.append(("pci", r["host"], r["pci_bus_id"]))
        for k in keys:
            if k in seen and seen[k] != r["rank"]:
                shared.append((seen[k], r["rank"], k[0]))
            seen.setdefault(k, r["rank"])
    return sorted(set(shared))


if __name__ == "__main__":
    main()
