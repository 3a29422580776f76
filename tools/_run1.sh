mkdir -p gpurun_out/r03g
exec > gpurun_out/r03g/log.txt 2>&1
timeout 600 python3 -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -5
for wl in c5 c2a; do python3 tools/kbench.py --workload $wl --rounds 2 --iters 10 --knobs "32=1,2"; done
python3 tools/kbench.py --shape 128,512,28,28 --active 1 --rounds 2 --iters 10 --knobs "32=1,2"
python3 tools/kbench.py --shape 128,256,56,56 --dtype bfloat16 --rounds 2 --iters 10 --knobs "32=1,2"
python3 tools/kbench.py --shape 128,256,56,56 --dtype bfloat16 --active 1 --rounds 2 --iters 10 --knobs "32=1,2"
python3 tools/kbench.py --shape 64,128,112,112 --active 1 --rounds 2 --iters 10 --knobs "32=1,2"
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
