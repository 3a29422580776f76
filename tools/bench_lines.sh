#!/usr/bin/env bash
# bench.py lines of several workloads, one summary line each (GPU box):  bash tools/bench_lines.sh <outdir> wl1 wl2 ...
OUT=$1; shift
mkdir -p $OUT
for wl in "$@"; do
    python3 bench.py --workload $wl --no-cpu-baseline --no-configs > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
    python3 - $OUT/bench_$wl.json $wl <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-8s ms/step %.4f  %s" % (sys.argv[2], j["ms_per_step"], "  ".join("%s %.4f ms %.2f (box %s)" % (k, v["ms"], v["GB/s"] / 8000, ("%.2f" % v["frac_of_box"]) if "frac_of_box" in v else "-") for k, v in j["kernels"].items())))
except Exception as e:
    print("bench failed", sys.argv[2], e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done
