#!/bin/bash
# L2 / fabric counters of K1 with and without a brick schedule (run on the GPU box)
OUT=$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  BRICKS=1 BRICK_ONE="$2" rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- /tmp/spmm_bench 256 64 64 2 > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pad8' not in r.get('Kernel_Name',''): continue
        acc[(r['Grid_Size'],r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for (k,c),v in sorted(acc.items()):
        g.write("grid=%-10s %-34s mean=%.6g n=%d\n"%(k,c,sum(v)/len(v),len(v)))
print(open(out+'/summary.txt').read())
PY
