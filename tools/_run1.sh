mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
for seed in 7 8 9 10; do timeout 300 python3 tools/fuzz_round2.py --seconds 75 --seed $seed 2>&1 | tail -3 | cut -c1-600; done
timeout 600 python3 -m pytest tests/test_fuzz_families_gpu.py -x -q -m gpu 2>&1 | tail -3
