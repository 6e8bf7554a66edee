#!/bin/bash
# fabric/L2 counters of the pad8 kernel for a given column-pass width (run on the GPU box)
OUT=$1; PASS=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for grp in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum"; do
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p_$(echo $grp | cut -c1-12) -- /tmp/spmm_bench 256 64 64 2 4 4 1 $PASS > $OUT/log.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pad8' not in r.get('Kernel_Name',''): continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print("%-34s per-launch mean=%.6g launches=%d"%(c,sum(v)/len(v),len(v)))
PY
