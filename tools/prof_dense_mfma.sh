#!/bin/bash
# MFMA busy fraction of the Gram / panel-update kernels from PMC counters (run on the GPU box): tools/prof_dense_mfma.sh <outdir>
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -I $R/gcge_amd/csrc/hip -I $R/include $R/tools/dense_bench.hip $R/gcge_amd/csrc/hip/gram_mfma.hip $R/gcge_amd/csrc/hip/lincomb_mfma.hip $R/gcge_amd/csrc/hip/vec_kernels.hip -o $OUT/dense_bench || exit 1
$OUT/dense_bench > $OUT/plain.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $OUT/dense_bench > $OUT/log$i.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if not any(t in k for t in ('gram_tile','lincomb_kernel','mfma_peak')): continue
        acc[(k.split('(')[0][-40:],r['Counter_Name'])].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print("%-42s %-30s mean=%.6g launches=%d"%(c[0],c[1],sum(v)/len(v),len(v)))
PY
cat $OUT/plain.log
