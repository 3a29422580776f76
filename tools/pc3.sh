#!/bin/bash
export TMPDIR=/tmp
for lib in D1 NOSTAGE NOCOMP; do
  export SHIFTND_HIP_LIB=$PWD/variants/$lib.so
  OUT=gpurun_out/pc_$lib; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python3 tools/kbench.py --workload c3 --rounds 2 --iters 5 > $OUT/log1.txt 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- python3 tools/kbench.py --workload c3 --rounds 2 --iters 5 > $OUT/log2.txt 2>&1
  python3 - $OUT $lib <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root=sys.argv[1]
res=defaultdict(dict); dur=defaultdict(list)
for f in sorted(glob.glob(os.path.join(root,'pmc*','**','*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'true, true, 3' not in k: continue
        res[r['Counter_Name']].setdefault('v',[]).append(float(r['Counter_Value']))
for f in sorted(glob.glob(os.path.join(root,'pmc1','**','*kernel_trace.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'true, true, 3' in r['Kernel_Name']: dur['d'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print('==', sys.argv[2], 'avg us %.1f' % (sum(dur['d'])/len(dur['d'])/1e3))
for k,v in sorted(res.items()): print('   %-24s %14.0f' % (k, sum(v['v'])/len(v['v'])))
PY
done
