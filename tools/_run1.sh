mkdir -p gpurun_out/r03v
exec > gpurun_out/r03v/log3.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -8
python3 bench.py --workload c2a --no-cpu-baseline --steps 20 | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2a', j['ms_per_step'], {k:(round(v['ms'],4), round(v.get('frac_of_box',0),3)) for k,v in j['kernels'].items()})"
