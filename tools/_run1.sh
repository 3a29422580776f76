mkdir -p gpurun_out/r03x
exec > gpurun_out/r03x/log.txt 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for wl in c2 c3 c4 c5; do
  bash profiles/collect.sh $wl --workload $wl > gpurun_out/r03x/collect_$wl.log 2>&1
  cp gpurun_out/prof_$wl/summary.txt gpurun_out/r03x/${wl}_summary.txt; cp gpurun_out/prof_$wl/traffic.json gpurun_out/r03x/${wl}_traffic.json
done
python3 bench.py > gpurun_out/r03x/bench_c2.json 2> gpurun_out/r03x/bench_c2.err
for wl in c3 c4 c5 c2a; do python3 bench.py --workload $wl --no-cpu-baseline > gpurun_out/r03x/bench_$wl.json 2> gpurun_out/r03x/bench_$wl.err; done
for pad in 1 2 3 4; do python3 bench.py --pad $pad --no-cpu-baseline > gpurun_out/r03x/bench_c2_pad$pad.json 2>/dev/null; done
python3 bench.py --gpus 2 --allow-oversubscribe --no-cpu-baseline --steps 10 > gpurun_out/r03x/bench_gpus2.json 2> gpurun_out/r03x/bench_gpus2.err
