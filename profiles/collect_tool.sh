#!/usr/bin/env bash
# rocprofv3 evidence for a tools/ script (kernels outside bench.py's workloads: channels-last, ragged rows, ...):
#   bash profiles/collect_tool.sh <tag> <script.py> [args...]
# kernel trace + FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/prof_<tag>/summary.txt (copy what should be judged to profiles/).
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$@" > "$OUT/run_trace.log" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -o p -- python3 "$@" > "$OUT/run_pmc_$ctr.log" 2>&1 || echo "pmc pass $ctr failed"
done
python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
