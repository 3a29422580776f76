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
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_self_launch_world2_cpu():
    r = _run(["--gpus", "2", "--device", "cpu", "--shape", "4,8,16,16", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    j = lines[0]
    ranks = dict(j["config"]["ranks"])
    per_rank = ranks.pop("per_rank")
    assert j["n_gpus"] == 2 and ranks == {"world": 2, "backend": "gloo", "devices": 0, "oversubscribed": False, "device_name": None}
    # every rank describes itself (all-gathered): rank ids, process-group size and backend as the process group reports them
    assert [r["rank"] for r in per_rank] == [0, 1] and len(set(r["pid"] for r in per_rank)) == 2
    assert all(r["pg_world_size"] == 2 and r["pg_backend"] == "gloo" and r["device_index"] is None for r in per_rank)
    assert len(j["per_rank_ms"]) == 2 and abs(max(j["per_rank_ms"]) - j["ms_per_step"]) < 1e-9
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["warmup"] == 1
    assert "cpu_baseline" not in j and j["roofline"] is None
    assert "not a measurement" in j["data"]
    # weak scaling: value counts the elements of both ranks
    assert abs(j["value"] - 2 * 4 * 8 * 16 * 16 / (j["ms_per_step"] * 1e-3) / 1e9) < 1e-9


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
    assert len(j[0]["per_rank_ms"]) == 1 and abs(j[0]["per_rank_ms"][0] - j[0]["ms_per_step"]) < 1e-9
    # the rank describes the device RCCL bound it to (what the first 8-GPU run will be read by)
    pr = j[0]["config"]["ranks"]["per_rank"]
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
