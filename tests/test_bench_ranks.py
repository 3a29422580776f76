"""bench.py's own rank code: the self-launcher (`--gpus N` without WORLD_SIZE starts N rank
processes before anything touches a GPU), init_process_group / barrier / all_reduce(MAX) / the one JSON line of rank 0.
On this CPU container the ranks run the CPU dispatch key over gloo (`--device cpu`, a functional check that the JSON
line labels as such); on the GPU box the same launcher runs 2 ranks on the one GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _json_lines(out):
    """the RESULT lines of a run (rank 0 also prints one `{"bench_detail": ...}` line before its RESULT line)"""
    return [j for j in (json.loads(l) for l in out.splitlines() if l.startswith("{")) if "bench_detail" not in j]


def _detail(out):
    d = [j for j in (json.loads(l) for l in out.splitlines() if l.startswith("{")) if "bench_detail" in j]
    assert len(d) == 1
    return d[0]["bench_detail"]


def _last_line_is_the_result(out):
    last = out.rstrip("\n").splitlines()[-1]
    j = json.loads(last)
    sys.path.insert(0, ROOT)
    import bench
    assert "bench_detail" not in j and "metric" in j and len(last) <= bench.HEADLINE_MAX_BYTES < 4096
    return j


def test_self_launch_world2_cpu():
    r = _run(["--gpus", "2", "--device", "cpu", "--shape", "4,8,16,16", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    j = lines[0]
    assert _last_line_is_the_result(r.stdout) == j
    ranks = dict(j["config"]["ranks"])
    assert [pr[0] for pr in ranks.pop("per_rank")] == [0, 1]   # compact in the RESULT line: [rank, host, device index, pci id]
    assert j["n_gpus"] == 2 and ranks == {"world": 2, "backend": "gloo", "devices": 0, "oversubscribed": False, "device_name": None}
    per_rank = _detail(r.stdout)["config"]["ranks"]["per_rank"]
    # every rank describes itself (all-gathered): rank ids, process-group size and backend as the process group reports them
    assert [r["rank"] for r in per_rank] == [0, 1] and len(set(r["pid"] for r in per_rank)) == 2
    assert all(r["pg_world_size"] == 2 and r["pg_backend"] == "gloo" and r["device_index"] is None for r in per_rank)
    assert len(j["per_rank_ms"]) == 2 and abs(max(j["per_rank_ms"]) - j["ms_per_step"]) < 1e-4 * j["ms_per_step"]
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["warmup"] == 1
    assert "cpu_baseline" not in j and j["roofline"] is None
    assert "not a measurement" in j["data"]
    # weak scaling: value counts the elements of both ranks
    d = _detail(r.stdout)
    assert abs(d["value"] - 2 * 4 * 8 * 16 * 16 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-9
    assert abs(j["value"] - d["value"]) <= 1e-4 * d["value"]   # (the RESULT line carries 5 significant digits)


def test_under_a_launcher_env_world2_cpu():
    """what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` does: ranks from the env"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    args = ["--gpus", "2", "--device", "cpu", "--workload", "c3", "--shape", "2,4,4,6,8", "--steps", "1", "--warmup", "0"]
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert len(_json_lines(outs[0][0])) == 1 and _json_lines(outs[1][0]) == []
    assert _json_lines(outs[0][0])[0]["n_gpus"] == 2


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "4", "--device", "cpu", "--shape", "4,8,16,16"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_dying_rank_stops_the_job_quickly():
    """rank 1 dies after the rendezvous: the launcher stops rank 0 (which would otherwise wait in its first barrier for the
    collective timeout) and exits non-zero within seconds"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--device", "cpu", "--shape", "4,8,16,16", "--steps", "2", "--warmup", "1"],
             env={"SHIFTND_BENCH_FAIL_RANK": "1", "SHIFTND_BENCH_TIMEOUT_S": "600"}, timeout=300)
    assert r.returncode != 0 and "the other ranks were stopped" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 120  # (most of it is two interpreters importing torch)
    assert _json_lines(r.stdout) == []


def test_no_cpu_fallback_for_the_measured_path():
    """default device is the GPU; without one the bench refuses instead of timing the CPU key"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run(["--steps", "1", "--no-cpu-baseline"])
    assert r.returncode != 0 and "needs an MI355X" in r.stderr


@pytest.mark.gpu
def test_self_launch_two_ranks_on_this_gpu_box():
    """`python bench.py --gpus 2` on a 1-GPU box: both ranks share the GPU and rendezvous over gloo (with >= 2 GPUs:
    RCCL); functional check of the launcher + rank code with the HIP kernels"""
    import torch
    if torch.cuda.device_count() < 2:
        refused = _run(["--gpus", "2", "--shape", "8,32,56,56", "--steps", "1", "--warmup", "0"], timeout=120)
        assert refused.returncode != 0 and "--allow-oversubscribe" in refused.stderr  # never a silent degradation
    r = _run(["--gpus", "2", "--shape", "8,32,56,56", "--steps", "3", "--warmup", "1", "--allow-oversubscribe"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1 and j[0]["n_gpus"] == 2
    assert j[0]["roofline"]["kernel"].startswith(("plane_", "sweep_", "step_"))
    assert "cpu_baseline" not in j[0]


@pytest.mark.gpu
def test_rccl_calls_of_the_rank_code_on_one_gpu():
    """the multi-GPU path's collectives with the backend it uses on a real node -- RCCL (`nccl`), device-bound process
    group, barrier, all_gather and all_reduce(MAX) of GPU tensors -- on the one GPU of this box (world size 1)"""
    r = _run(["--gpus", "1", "--force-process-group", "--shape", "8,32,56,56", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-probe"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1 and j[0]["n_gpus"] == 1 and j[0]["config"]["ranks"]["backend"] == "nccl"
    assert len(j[0]["per_rank_ms"]) == 1 and abs(j[0]["per_rank_ms"][0] - j[0]["ms_per_step"]) < 1e-4 * j[0]["ms_per_step"]
    assert _last_line_is_the_result(r.stdout) == j[0]
    # the rank describes the device RCCL bound it to (what the first 8-GPU run will be read by)
    assert j[0]["config"]["ranks"]["per_rank"][0][2] == 0
    pr = _detail(r.stdout)["config"]["ranks"]["per_rank"]
    assert len(pr) == 1 and pr[0]["device_index"] == 0 and pr[0]["pg_backend"] == "nccl" and pr[0]["pg_world_size"] == 1
    assert pr[0]["device_name"] and pr[0]["pci_bus_id"] and pr[0]["hbm_bytes"] > 2 ** 37


def test_distinct_device_guard_only_fires_on_provable_sharing():
    """a torch build that reports no uuid / an all-zero PCI id must not make eight ranks on eight devices look
    like one GPU; two ranks on one device index of one host (or with one real uuid) must still abort."""
    sys.path.insert(0, ROOT)
    import bench
    def rk(i, dev, uuid=None, pci=None, host="h"):
        return {"rank": i, "host": host, "device_index": dev, "uuid": uuid, "pci_bus_id": pci}
    assert bench.distinct_device_conflicts([rk(i, i, "00000000-0000-0000-0000-000000000000", "0000:00:00") for i in range(8)]) == []
    assert bench.distinct_device_conflicts([rk(i, i) for i in range(8)]) == []
    assert bench.distinct_device_conflicts([rk(i, i, "6462-%d" % i, "0000:%02x:00" % (0x10 + i)) for i in range(8)]) == []
    assert bench.distinct_device_conflicts([rk(0, 0), rk(1, 0)]) == [(0, 1, "index")]
    assert bench.distinct_device_conflicts([rk(0, 0, "abc"), rk(1, 1, "abc")]) == [(0, 1, "uuid")]
    assert bench.distinct_device_conflicts([rk(0, 0, None, "0000:72:00"), rk(1, 1, None, "0000:72:00")]) == [(0, 1, "pci")]
    assert bench.distinct_device_conflicts([rk(0, 0, host="a"), rk(1, 0, host="b")]) == []


def _worst_case_record(n_ranks=8):
    """a full record shaped like the default GPU run's (every config present, long kernel names, 8 ranks, the whole
    cpu_baseline record of oracle/ref_bench.py --both --full) -- what the RESULT line has to be cut from"""
    sys.path.insert(0, ROOT)
    import bench
    kern = {"ms": 1.5207577705383302, "median_ms": 1.5204139947891235, "min_ms": 1.4988930225372314, "GB/s": 6486.8996227505095,
            "stream": "2R1W", "frac_of_box": 1.0145767900825047}
    rl = {"bound": "hbm", "kernel": "cl_tiled_active_forward_3d_ncdhw_grad", "achieved": 6486.8996227505095, "peak": 8000.0, "unit": "GB/s",
          "frac": 0.8108624528438136, "traffic": 9860488542.315788, "traffic_source": "profiles/r06_c2_traffic.json (round 6)",
          "avg_kernel_ms": 1.5207577705383302, "algorithmic_bytes": 9865003008, "frac_of_box": 1.0145767900825047,
          "box_stream": {"probe": "tools/stream_probe", "buffer_bytes": 3288334336, "1R1W_GBps": 6471.1, "1R1W_shape": "K1 nt",
                         "2R1W_GBps": 6393.7, "2R1W_shape": "K1 nt", "read_GBps": 6489.9, "write_GBps": 5894.9, "device": "", "cus": 256,
                         "clock_mhz": 2400}}
    entry = {"value": 0.20902466285248827, "unit": "Gelem/s", "cores": 256, "kind": "reference",
             "sample": "Shift2d SSL fwd+bwd N24 C256 224x224 fp32 pad 0, best of 2 (fwd 268.5 ms, bwd 1206.3 ms)"}
    base = dict(entry, host_cpu="AMD EPYC 9575F 64-Core Processor", host_cores=256, usable_cores=256, single_thread=dict(entry, cores=1),
                note="x" * 160, n4_slice=[entry, entry], full_size=[dict(entry, cores=1), entry],
                own_cpu_key={"value": 0.31, "unit": "Gelem/s", "cores": 256, "kind": "port", "sample": "y" * 90})
    per_rank = [{"rank": i, "local_rank": i, "host": "mi355x-node-0123456789", "pid": 1000 + i, "device_index": i, "world_size": n_ranks,
                 "backend": "nccl", "device_name": "AMD Instinct MI355X", "uuid": "64623238-3930-6435-3463-61653034613%d" % i,
                 "pci_bus_id": "0000:%02x:00" % (0x10 + i), "hbm_bytes": 309220868096, "pg_world_size": n_ranks, "pg_backend": "nccl"}
                for i in range(n_ranks)]
    configs = {name: {"workload": bench.WORKLOADS[w][4] + ", padding %d" % p, "steps": 20, "warmup": 5, "ms_per_step": 1.5923029499390395,
                      "value": 129.07147851975546, "unit": "Gelem/s", "dtype": "f32", "achieved_hbm_GBps_step": 2581.429570395109,
                      "kernels": {"cl_tiled_active_forward_3d": kern, "cl_tiled_backward_3d_ncdhw_grad": kern},
                      "roofline": {k: rl[k] for k in ("kernel", "frac", "frac_of_box", "traffic", "traffic_source", "avg_kernel_ms",
                                                      "algorithmic_bytes")}}
               for name, w, p in bench.EXTRA_CONFIGS}
    return {"metric": "Gelem/s, Shift2d fwd+bwd N64/C256/224x224", "value": 320.1259813021249, "unit": "Gelem/s", "n_gpus": n_ranks,
            "steps": 20, "warmup": 5, "ms_per_step": 2.5680001999717206, "ms_per_step_median": 2.5842190079856664,
            "ms_per_step_min": 2.568166994024068, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": bench.WORKLOADS["c2"][4] + ", padding 0, per GPU; batch sharded over 8 GPU(s), no collectives",
                       "path": "torch.ops.torchshifts._shift2d_forward/_backward -> libshiftnd_hip.so (step kernels)",
                       "ranks": {"world": n_ranks, "backend": "nccl", "devices": 8, "oversubscribed": False,
                                 "device_name": "AMD Instinct MI355X", "per_rank": per_rank}},
            "per_rank_ms": [2.5680001999717206] * n_ranks, "achieved_hbm_GBps_step": 6402.519626042498,
            "kernels": {"step_gather_forward": kern, "step_backward": kern}, "roofline": rl, "cpu_baseline": base,
            "configs": configs, "configs_wall_s": 6.563866232998407,
            "fallback_tail": {"backward": [164, 5131], "forward": [136, 5131], "forward_quantized": [74, 869],
                              "source": "profiles/r06_route_census.txt"}}


def test_result_line_is_last_and_bounded():
    """round-5 verdict: the RESULT line must survive a consumer that keeps a few KB of the tail of stdout.  The full record goes
    out first (`bench_detail`), the RESULT line last, under HEADLINE_MAX_BYTES, with everything the contract and SURVEY 8d name."""
    import io
    sys.path.insert(0, ROOT)
    import bench
    rec = _worst_case_record()
    buf = io.StringIO()
    bench.emit(rec, out=buf)
    lines = buf.getvalue().rstrip("\n").split("\n")
    assert len(lines) == 2 and json.loads(lines[0])["bench_detail"]["configs"].keys() == rec["configs"].keys()
    last = lines[-1]
    assert len(last) <= bench.HEADLINE_MAX_BYTES < 4096
    j = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in j, k
    assert j["config"]["workload"].startswith("Shift2d SSL fwd+bwd N64 C256 224x224 fp32")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k
    for k in ("value", "unit", "cores", "kind"):
        assert j["cpu_baseline"][k] is not None
    assert j["cpu_baseline"]["own_cpu_key"]["value"] == 0.31 and j["cpu_baseline"]["single_thread"]["cores"] == 1
    # every BASELINE config beyond the headline is in the compact block: [ms_per_step, dominant kernel, frac, traffic ratio]
    for name in bench.BASELINE_CONFIGS:
        ms, kernel, frac, ratio = j["configs"][name]
        assert ms > 0 and isinstance(kernel, str) and 0 < frac < 1.2 and (ratio is None or ratio > 0.5)
    assert set(bench.BASELINE_CONFIGS) == {"c2_pad1", "c2_pad2", "c2_pad3", "c2_pad4", "c3_pad0", "c3_pad1", "c3_pad2", "c3_pad3",
                                           "c3_pad4", "c4", "c5"}
    # the driver's view: the last 8 KB of stdout + a stderr banner still hold the whole RESULT line
    tail = (buf.getvalue() + "\n---- stderr ----\nwarning\n")[-8192:]
    assert last in tail
    # a failed / skipped config is still named
    rec["configs"]["c4"] = {"error": "RuntimeError('x')"}
    assert json.loads(bench.headline_line(rec))["configs"]["c4"][1] == "error"
    # and a smaller limit sheds optional parts, never the contract's
    small = json.loads(bench.headline_line(rec, limit=2200))
    assert "roofline" in small and "cpu_baseline" in small and small["value"] == j["value"]
