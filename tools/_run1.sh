mkdir -p gpurun_out/r03z
exec > gpurun_out/r03z/log.txt 2>&1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for sh in 128,512,56,56 128,1024,28,28 128,2048,14,14 128,2048,7,7 64,256,112,112; do
python3 tools/pool_bench.py --quant --shape $sh --pool 2 --iters 50 2>&1 | tail -3
done
