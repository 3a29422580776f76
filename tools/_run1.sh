mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "3d_walk_forward" 2>&1 | tail -3
python3 tools/kbench.py --workload c3 --rounds 4 --iters 10 --libs tree,nopk 2>&1 | tail -6
python3 tools/kbench.py --workload c3f --knobs "35=32" --rounds 4 --iters 10 --libs tree,nopk 2>&1 | tail -6
