mkdir -p gpurun_out/r03g
exec > gpurun_out/r03g/log2.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "FAILED|passed|failed" | sed 's/\[.*//' | sort | uniq -c
