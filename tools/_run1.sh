mkdir -p gpurun_out/r03u
exec > gpurun_out/r03u/log.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py tests/test_fuzz_families_gpu.py -x -q 2>&1 | tail -5
for wl in c2 c5 c2a c5a; do python3 tools/kbench.py --workload $wl --rounds 2 --iters 10 --knobs "35=2,4" | grep bwd; done
python3 tools/kbench.py --workload c5 --pad 3 --rounds 2 --iters 10 --knobs "35=2,4" | grep bwd
python3 tools/kbench.py --shape 128,256,56,56 --dtype bfloat16 --rounds 2 --iters 20 --knobs "35=2,4" | grep bwd
python3 tools/kbench.py --shape 128,256,56,56 --active 1 --rounds 2 --iters 20 --knobs "35=2,4" | grep bwd
python3 tools/kbench.py --shape 128,512,28,28 --active 1 --rounds 2 --iters 20 --knobs "35=2,4" | grep bwd
python3 tools/kbench.py --shape 128,512,28,28 --rounds 2 --iters 20 --knobs "35=2,4" | grep bwd
