#!/usr/bin/env bash
# round-4 evidence: rocprofv3 kernel trace + PMC passes (profiles/collect.sh) and a plain bench line for every workload;
# results land in gpurun_out/r04p/ as r04_<tag>_{rocprof_summary.txt,traffic.json,bench.json} (copy them to profiles/)
OUT=gpurun_out/r04p
mkdir -p $OUT
run() {  # tag, bench args...
    tag=$1; shift
    bash profiles/collect.sh $tag "$@" > $OUT/collect_$tag.log 2>&1
    cp gpurun_out/prof_$tag/summary.txt $OUT/r04_${tag}_rocprof_summary.txt 2>/dev/null
    cp gpurun_out/prof_$tag/traffic.json $OUT/r04_${tag}_traffic.json 2>/dev/null
    python3 bench.py "$@" --no-cpu-baseline > $OUT/r04_${tag}_bench.json 2> $OUT/bench_$tag.err
}
for wl in "$@"; do
    case $wl in
        c3_pad*) run $wl --workload c3 --pad ${wl#c3_pad} ;;
        *) run $wl --workload $wl ;;
    esac
done
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/r04_*_bench.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "ms/step %.3f" % j["ms_per_step"], {k: (round(v["ms"], 4), round(v["GB/s"]), round(v.get("frac_of_box", 0), 2)) for k, v in j["kernels"].items()})
    except Exception as e:
        print("bench failed", f, e)
PY
