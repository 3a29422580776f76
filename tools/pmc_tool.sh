#!/usr/bin/env bash
# usage: tools/pmc_tool.sh <outdir> <filter> <script.py> [args...]  -- instruction mix / waits / LDS conflicts per kernel of a tools/ script (GPU box)
OUT=$1; FILT=$2; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p -- python3 "$@" > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -o p -- python3 "$@" > $OUT/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o p -- python3 "$@" > $OUT/log3.txt 2>&1
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root, filt = sys.argv[1], sys.argv[2].split(",")
acc=defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root,'p*','**','*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('shiftnd::(anonymous namespace)::','').replace('void ','')[:80]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    if not any(t in k for t in filt): continue
    w=sum(c['SQ_WAVES'])/len(c['SQ_WAVES']) if 'SQ_WAVES' in c else 1
    print('==',k,'waves %d'%w)
    print('   per wave: ' + '  '.join('%s %.0f' % (n.replace('SQ_',''), (sum(v)/len(v))/w) for n,v in sorted(c.items()) if n!='SQ_WAVES'))
PY
