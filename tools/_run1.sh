mkdir -p gpurun_out/r03h
exec > gpurun_out/r03h/log.txt 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python3 bench.py ) > gpurun_out/r03h/bench_c2_default.json 2> gpurun_out/r03h/bench_c2_default.err
tail -4 gpurun_out/r03h/bench_c2_default.err
for wl in c3 c4 c5 c2a; do python3 bench.py --workload $wl --no-cpu-baseline --steps 20 > gpurun_out/r03h/bench_$wl.json 2> gpurun_out/r03h/bench_$wl.err; done
for pad in 1 2 3 4; do python3 bench.py --pad $pad --no-cpu-baseline --steps 20 > gpurun_out/r03h/bench_c2_pad$pad.json 2>/dev/null; done
bash profiles/collect.sh c2 --workload c2 > gpurun_out/r03h/collect_c2.log 2>&1
cp gpurun_out/prof_c2/summary.txt gpurun_out/r03h/c2_summary.txt; cp gpurun_out/prof_c2/traffic.json gpurun_out/r03h/c2_traffic.json
