#!/bin/bash
# PMC passes for the SpMM micro-benchmark (run on the GPU box via gpurun).
# usage: tools/prof_spmm.sh <outdir> <bench args...>
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES" \
 "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
 "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- /tmp/spmm_bench "$@" > $OUT/p$i.log 2>&1
done
# collapse: kernel name, counter, mean value over dispatches of spmm kernels
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if 'spmm' not in k and 'copy' not in k: continue
        short='spmm' if 'spmm' in k else 'copy'
        acc[(short,r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for (k,c),v in sorted(acc.items()):
        g.write("%-6s %-40s mean=%.6g n=%d\n"%(k,c,sum(v)/len(v),len(v)))
print(open(out+'/summary.txt').read())
PY
