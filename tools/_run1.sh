mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "3d" 2>&1 | tail -5
timeout 300 python3 tools/fuzz_round2.py --seconds 60 --seed 3 2>&1 | tail -3 | cut -c1-400
python3 tools/kbench.py --workload c3s --knobs "35=0,16" --rounds 3 --iters 10 2>&1 | tail -6
python3 tools/kbench.py --workload c3fs --knobs "35=0,16" --rounds 3 --iters 10 2>&1 | tail -6
