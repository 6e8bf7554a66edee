#!/bin/bash
# SQ / cache counters of the K1 kernels for matrices without a pattern form (spmm_tile.hip, spmm_dense.hip, spmm_pad8.hip)
# on the SiO2-like matrix.    tools/prof_tile.sh <outdir> [G K m]        (run on the GPU box; TILE_MODE / DENSE_MODE as tools/tile_probe.py)
#   PROBE=tools/offset_probe.py tools/prof_tile.sh <outdir> 256 64     (any probe script: its arguments follow the directory)
OUT=$GRAFT_REPO_ROOT/$1; shift; PROBE=${PROBE:-tools/tile_probe.py}; ARGS="${@:-96 354 64}"
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/$PROBE $ARGS > $OUT/log$i.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/$PROBE $ARGS > $OUT/log_trace.txt 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if 'spmm_star2_kernel' in k:
            kk=k.replace(' ','')
            tag='star' + ('+sums' if 'kernel<true' in kk else '') + (' 16col' if ',8,0,' in kk else ' 8col') + (' masked' if kk.split('>')[0].endswith('true') else '')
        elif 'spmm_star3_kernel' in k: tag='star3' + ('+sums' if 'kernel<true' in k.replace(' ','') else '')
        elif 'spmm_star_kernel' in k: tag='star'
        elif 'spmm_tile_kernel' in k: tag='tile'
        elif 'spmm_pad8' in k: tag='pad8'
        elif 'spmm_dense' in k: tag='dense'
        elif 'spmm_pattern_chain2' in k: tag='chain2' + ('+values' if 'true>' in k.replace(' ', '') else '')
        elif 'spmm_pattern' in k: tag='pattern'
        else: continue
        acc[(tag,r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for c,v in sorted(acc.items()):
        line="%-14s %-34s per-launch mean=%.6g launches=%d"%(c[0],c[1],sum(v)/len(v),len(v))
        print(line); fo.write(line+"\n")
    for f in glob.glob(out+'/trace/**/*kernel_stats.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'spmm' in r.get('Name',''):
                line="stats %s calls=%s avg_ns=%s"%(r['Name'][:60],r.get('Calls'),r.get('AverageNs'))
                print(line); fo.write(line+"\n")
PY
