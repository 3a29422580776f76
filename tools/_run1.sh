mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 600 python3 tools/cl_tiled_bench.py 2>&1 | tail -12
