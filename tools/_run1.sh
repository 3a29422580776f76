mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "3d_walk" 2>&1 | tail -4
python3 tools/kbench.py --workload c3 --knobs "35=0,16" --rounds 3 --iters 10 2>&1 | tail -6
python3 tools/kbench.py --workload c3f --knobs "35=0,16" --rounds 3 --iters 10 2>&1 | tail -6
