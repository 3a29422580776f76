mkdir -p gpurun_out/r03k
exec > gpurun_out/r03k/log.txt 2>&1
timeout 900 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "FAILED|passed|failed" | sed 's/\[.*//' | sort | uniq -c
cp activesparseshifts-pytorch_amd/torchshifts/libshiftnd_hip.so variants/new.so
for wl in c2 c5 c2a c5a; do python3 tools/kbench.py --workload $wl --rounds 2 --iters 10 --libs base2,new; done
