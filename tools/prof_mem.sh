#!/usr/bin/env bash
# usage: tools/prof_mem.sh <outdir> <python script + args...>   (GPU box): FETCH/WRITE/L2 counters per kernel
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo "$ctr" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -o p -- python3 "$@" > $OUT/log_$name.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root=sys.argv[1]
acc=defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root,'*','**','*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('shiftnd::(anonymous namespace)::','').replace('void ','').replace('shiftnd::','')[:52]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    if not any(t in k for t in ('sweep_','plane_','strided_')): continue
    print('==',k)
    for name,v in sorted(c.items()):
        avg=sum(v)/len(v)
        extra=''
        if name=='FETCH_SIZE': extra=' -> %.3f GB read (x2 gfx950 correction)'%(avg*2*1024/1e9)
        if name=='WRITE_SIZE': extra=' -> %.3f GB written'%(avg*1024/1e9)
        print('   %-32s %16.0f%s' % (name, avg, extra))
PY
