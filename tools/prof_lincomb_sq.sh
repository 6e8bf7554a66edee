#!/bin/bash
# Where do the waves of the panel-update kernels spend their cycles?  (run on the GPU box): tools/prof_lincomb_sq.sh <outdir>
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for rf in ${LC_LIST:-0}; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    LC_RF=$rf rocprofv3 --pmc $grp --output-format csv -d $OUT/rf${rf}_p$i -- $GRAFT_REPO_ROOT/tools/_bin/dense_bench > $OUT/rf${rf}_log$i.txt 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,os
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/rf*_p*/**/*counter_collection.csv',recursive=True):
    tag=f[len(out)+1:].split('_')[0]
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if ('lincomb' not in k and 'gram_tile' not in k) or 'pad_c' in k: continue
        acc[(tag, k.split('(')[0][-44:], r['Counter_Name'])].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print("%-4s %-46s %-30s mean=%.6g launches=%d"%(c[0],c[1],c[2],sum(v)/len(v),len(v)))
PY
