mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
python3 tools/kbench.py --workload c5 --rounds 4 --iters 10 --libs tree,nopk 2>&1 | tail -6
python3 tools/kbench.py --workload c5 --pad 3 --rounds 4 --iters 10 --libs tree,nopk 2>&1 | tail -6
python3 tools/kbench.py --shape 128,256,56,56 --dtype bfloat16 --rounds 4 --iters 20 --libs tree,nopk 2>&1 | tail -6
