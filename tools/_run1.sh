mkdir -p gpurun_out/r03f
bash tools/gpu_round.sh r03f c2 c3 c4 c5 > gpurun_out/r03f/round.log 2>&1
python3 bench.py --workload c2a --no-cpu-baseline > gpurun_out/r03f/bench_c2a.json 2> gpurun_out/r03f/bench_c2a.err
bash profiles/collect_tool.sh extra tools/prof_workloads.py --iters 5 > gpurun_out/r03f/collect_extra.log 2>&1
cp gpurun_out/prof_extra/summary.txt gpurun_out/r03f/extra_summary.txt
