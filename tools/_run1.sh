mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 1200 python3 -m pytest tests/test_pooled_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -12
python3 tools/pool_bench.py --dtype bfloat16 --shape 8,128,16,112,112 --active 1 2>&1 | tail -6
python3 tools/pool_bench.py --dtype float32 --shape 8,128,16,112,112 --active 1 2>&1 | tail -6
