#!/usr/bin/env bash
# usage: tools/prof_counters.sh <outdir> <python script + args...>   (GPU box; PMC passes only, see profiles/collect.sh)
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o p -- python3 "$@" > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc2 -o p -- python3 "$@" > $OUT/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc3 -o p -- python3 "$@" > $OUT/log3.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root=sys.argv[1]
for f in sorted(glob.glob(os.path.join(root,'pmc*','**','*counter_collection.csv'), recursive=True)):
    acc=defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('shiftnd::(anonymous namespace)::','').replace('void ','')[:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in acc.items():
        if not any(t in k for t in ('sweep_','plane_','strided_','slide_')): continue
        print('==',k)
        for name,v in sorted(c.items()):
            print('   %-24s %16.0f' % (name, sum(v)/len(v)))
PY
