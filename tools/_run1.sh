mkdir -p gpurun_out/r03t
exec > gpurun_out/r03t/log4.txt 2>&1
python3 tools/_dbg.py 2>&1 | grep mismatches
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -6
