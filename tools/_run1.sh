mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 1200 python3 -m pytest tests/test_slide_gpu.py tests/test_hip_parity.py tests/test_fuzz_families_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -12
