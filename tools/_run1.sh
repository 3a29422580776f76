mkdir -p gpurun_out/r03n
exec > gpurun_out/r03n/log.txt 2>&1
timeout 600 python3 -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -8
python3 tools/kbench.py --workload c3 --rounds 2 --iters 20 --knobs "35=0,1"
python3 tools/kbench.py --workload c3f --rounds 2 --iters 20 --knobs "35=0,1"
python3 tools/kbench.py --workload c3s --rounds 2 --iters 20 --knobs "35=0,1"
python3 tools/kbench.py --workload c3 --pad 3 --rounds 2 --iters 20 --knobs "35=0,1"
