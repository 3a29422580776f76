#!/bin/bash
# Which kernels does the GPU suite launch?  rocprofv3 kernel trace of the whole `-m gpu` suite (minus the tests that start
# child processes); gpurun_out/suite_kernels/ keeps the per-kernel stats, tools/unlaunched_kernels.py compares them with the
# symbols of the library.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/suite_kernels
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/suite_kernels/raw -o suite -- python3 -m pytest tests -m gpu -q -x \
    --deselect tests/test_bench_ranks.py -p no:cacheprovider > gpurun_out/suite_kernels/pytest.log 2>&1
echo "pytest rc $?"
tail -3 gpurun_out/suite_kernels/pytest.log
f=$(find gpurun_out/suite_kernels/raw -name '*kernel_stats.csv' | head -1)
echo "stats: $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open("gpurun_out/suite_kernels/launched.txt", "w") as f:
    for r in rows:
        f.write("%s\t%s\n" % (r["Calls"], r["Name"]))
print(len(rows), "distinct kernels launched")
PY
rm -rf gpurun_out/suite_kernels/raw
