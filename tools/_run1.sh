mkdir -p gpurun_out/r03l
exec > gpurun_out/r03l/log.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15
