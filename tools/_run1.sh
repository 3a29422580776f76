mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "3d_walk_backward" 2>&1 | tail -3
python3 tools/kbench.py --workload c3 --knobs "38=0,8,4,2" --rounds 3 --iters 10 2>&1 | grep -E "bwd|copy"
python3 tools/kbench.py --workload c3s --knobs "38=0,8,4,2" --rounds 3 --iters 10 2>&1 | grep -E "bwd|copy"
