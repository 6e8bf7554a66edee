#!/bin/bash
# fabric counters of the CG-pass kernels and the plain product (run on the GPU box): tools/prof_cg_passes.sh <outdir> [N] [m]
OUT=$GRAFT_REPO_ROOT/$1; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/cg_pass_probe.py "$@" > $OUT/log$i.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,re
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        m=re.search(r'spmm_pattern_chain2_kernel<(\d+), (\d+), (\d+)>',k)
        if not m: continue
        acc[("chain2 LT=%s MODE=%s NW=%s"%m.groups(),r['Counter_Name'])].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print("%-32s %-30s per 16-column launch mean=%.6g launches=%d"%(c[0],c[1],sum(v)/len(v),len(v)))
PY
cat $OUT/log1.txt | tail -2
