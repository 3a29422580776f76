#!/usr/bin/env bash
TAG=${1:-r04b}
mkdir -p gpurun_out/$TAG
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "walk_backward" > gpurun_out/$TAG/pytest_walk.log 2>&1; tail -15 gpurun_out/$TAG/pytest_walk.log
for pad in 0 1 3; do
  timeout 300 python3 tools/kbench.py --workload c3 --pad $pad --rounds 3 --iters 20 2>&1 | grep -E "bwd|fwd|copy" | sed "s/^/c3 pad$pad /"
done
timeout 300 python3 tools/kbench.py --workload c3s --pad 0 --rounds 3 --iters 20 2>&1 | grep -E "bwd" | sed "s/^/c3s pad0 /"
timeout 300 python3 tools/kbench.py --workload c3 --pad 0 --rounds 3 --iters 20 --knobs "35=64" 2>&1 | grep -E "bwd" | sed "s/^/c3 old /"
