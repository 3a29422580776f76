mkdir -p gpurun_out/r03y
python3 tools/cl_tiled_bench.py 0 4 8 16 32 64 > gpurun_out/r03y/cl.txt 2>&1
python3 tools/cl_bench.py >> gpurun_out/r03y/cl.txt 2>&1
