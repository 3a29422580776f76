mkdir -p gpurun_out/r03s
exec > gpurun_out/r03s/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py tests/test_small_gpu.py -x -q 2>&1 | tail -8
python3 tools/kbench.py --workload c2a --rounds 2 --iters 10 --knobs "33=1,0" | grep fwd
python3 tools/kbench.py --shape 64,128,112,112 --active 1 --rounds 2 --iters 20 --knobs "33=1,0" | grep fwd
python3 tools/kbench.py --shape 32,64,224,224 --dtype float64 --active 1 --rounds 2 --iters 10 --knobs "33=1,0" | grep fwd
python3 tools/kbench.py --shape 128,1024,14,14 --dtype quint8 --rounds 2 --iters 50 --knobs "16=1,0"
python3 tools/kbench.py --shape 128,2048,7,7 --dtype quint8 --rounds 2 --iters 50 --knobs "16=1,0"
python3 tools/kbench.py --shape 128,1024,14,14 --dtype quint8 --pad 3 --rounds 2 --iters 50 --knobs "16=1,0"
