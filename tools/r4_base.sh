#!/usr/bin/env bash
# round-4 baseline: the new bench workloads on whatever kernels serve them, C3 for every padding
TAG=${1:-r04a}
mkdir -p gpurun_out/$TAG
for wl in c2crop c2acrop t1 t1a c1d c1da c1dh; do
    timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-probe --steps 20 --warmup 5 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err
done
for pad in 0 1 2 3 4; do
    timeout 300 python3 bench.py --workload c3 --pad $pad --no-cpu-baseline --no-probe --steps 30 --warmup 5 > gpurun_out/$TAG/bench_c3_pad$pad.json 2> gpurun_out/$TAG/bench_c3_pad$pad.err
done
python3 - gpurun_out/$TAG <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "ms/step %.3f" % j["ms_per_step"], {k: (round(v["ms"], 4), round(v["GB/s"])) for k, v in j["kernels"].items()})
    except Exception as e:
        print("bench failed", f, e)
PY
