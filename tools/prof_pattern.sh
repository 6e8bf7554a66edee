#!/bin/bash
# fabric/L2 counters of the pattern SpMM kernel through the python probe (run on the GPU box)
# usage: tools/prof_pattern.sh <outdir> [probe args...]
OUT=$GRAFT_REPO_ROOT/$1; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/spmm_probe.py "$@" > $OUT/log$i.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if "spmm" not in k and "pad8" not in k: continue
        acc[(k[:48],r['Counter_Name'])].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print("%-50s %-30s per-launch mean=%.6g launches=%d"%(c[0],c[1],sum(v)/len(v),len(v)))
PY
