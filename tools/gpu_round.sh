#!/usr/bin/env bash
# One gpurun call: GPU tests, the bench lines of every workload, and (optionally) the rocprofv3 evidence.
#   bash tools/gpu_round.sh <tag> [profile workloads...]
TAG=${1:-run}; shift || true
mkdir -p gpurun_out/$TAG
python3 -m pytest tests -x -q -m gpu > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest_gpu.log
tail -3 gpurun_out/$TAG/pytest_gpu.log
for wl in c2 c3 c4 c5; do
    python3 bench.py --workload $wl $( [ $wl != c2 ] && echo --no-cpu-baseline ) > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err
    python3 - gpurun_out/$TAG/bench_$wl.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(j["config"]["workload"][:50], "ms/step %.3f" % j["ms_per_step"], {k: round(v["ms"], 4) for k, v in j["kernels"].items()}, "frac %.3f" % j["roofline"]["frac"])
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
done
for wl in "$@"; do
    bash profiles/collect.sh $wl --workload $wl > gpurun_out/$TAG/collect_$wl.log 2>&1
    mkdir -p gpurun_out/$TAG/prof_$wl
    cp gpurun_out/prof_$wl/summary.txt gpurun_out/prof_$wl/traffic.json gpurun_out/$TAG/prof_$wl/ 2>/dev/null
done
