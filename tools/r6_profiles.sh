#!/usr/bin/env bash
# Round 6: rocprofv3 evidence (kernel trace + FETCH / WRITE counters) and a bench line per workload (GPU box, through gpurun):
#   bash tools/r6_profiles.sh wl1 wl2 ...        ("c3:1" = workload c3 with --pad 1)
# -> gpurun_out/r6p/r06_<tag>_{rocprof_summary.txt,traffic.json,bench.json}; copy what should be judged into profiles/.
mkdir -p gpurun_out/r6p
for spec in "$@"; do
    wl=${spec%%:*}; pad=0; tag=$wl
    if [[ "$spec" == *:* ]]; then pad=${spec##*:}; tag=${wl}_pad$pad; fi
    bash profiles/collect.sh $tag --workload $wl --pad $pad > gpurun_out/r6p/collect_$tag.log 2>&1
    cp gpurun_out/prof_$tag/summary.txt gpurun_out/r6p/r06_${tag}_rocprof_summary.txt 2>/dev/null
    cp gpurun_out/prof_$tag/traffic.json gpurun_out/r6p/r06_${tag}_traffic.json 2>/dev/null
    python3 bench.py --workload $wl --pad $pad --no-cpu-baseline --no-configs 2> gpurun_out/r6p/bench_$tag.err | tail -1 > gpurun_out/r6p/r06_${tag}_bench.json
    python3 - gpurun_out/r6p/r06_${tag}_bench.json $tag <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read())
    print("%-10s ms/step %.4f  %s" % (sys.argv[2], j["ms_per_step"], "  ".join("%s %.4f ms %.2f (box %s)" % (k, v["ms"], v["GB/s"] / 8000, ("%.2f" % v["frac_of_box"]) if "frac_of_box" in v else "-") for k, v in j["kernels"].items())))
except Exception as e:
    print("bench failed", sys.argv[2], e)
PY
    rm -rf gpurun_out/prof_$tag
done
