mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --workload c3 2>&1 | tail -1 | cut -c1-900
