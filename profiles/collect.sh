#!/usr/bin/env bash
# Collect the rocprofv3 evidence for one bench.py workload on the GPU box (run through gpurun):
#   bash profiles/collect.sh <tag> [bench.py args...]
# Writes gpurun_out/prof_<tag>/ (scratch) and prints the per-kernel summaries; copy what should be
# judged into profiles/ (tracked).  PMC counters are collected in their own passes, without any
# tracing option besides --kernel-trace (gpurun refuses other combinations).
set -u
TAG=${1:-c2}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# (--no-configs: a profile holds ITS workload's kernels only -- round-5 verdict: one average over every workload of the default run hid
#  single instantiations)
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py $ARGS > "$OUT/bench_trace.log" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    name=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -o p -- python3 bench.py $ARGS > "$OUT/bench_pmc_$name.log" 2>&1 || echo "pmc pass $ctr failed"
done
python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
