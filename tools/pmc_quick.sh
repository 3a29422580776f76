#!/usr/bin/env bash
# usage: tools/pmc_quick.sh <outdir> <bench.py args...>   -- instruction mix and LDS conflicts per kernel (GPU box)
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-probe $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p -- python3 bench.py $ARGS > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -o p -- python3 bench.py $ARGS > $OUT/log2.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root=sys.argv[1]
acc=defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root,'p*','**','*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('shiftnd::(anonymous namespace)::','').replace('void ','')[:70]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    if not any(t in k for t in ('crop_','span_','step_back','step_forward','step_active','step_gather','walk_')): continue
    w=sum(c['SQ_WAVES'])/len(c['SQ_WAVES']) if 'SQ_WAVES' in c else 1
    print('==',k,'waves %d'%w)
    print('   per wave: ' + '  '.join('%s %.0f' % (n.replace('SQ_',''), (sum(v)/len(v))/w) for n,v in sorted(c.items()) if n!='SQ_WAVES'))
PY
