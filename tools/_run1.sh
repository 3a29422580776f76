mkdir -p gpurun_out/r03w
exec > gpurun_out/r03w/log2.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -8
