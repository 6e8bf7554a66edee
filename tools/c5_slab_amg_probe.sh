# config 5 with BlockAMG on 2 row slabs sharing one GPU (rehearsal): which switch decides whether it converges?
mkdir -p gpurun_out/r5b
run() { tag=$1; shift; envs=$1; shift
  env $envs timeout -k 10 120 python bench.py --config c5 --gpus 2 --rehearse --size 48 --atoms 40,2.0,5.0 --steps 1 --no-cpu --no-extra "$@" > gpurun_out/r5b/c5s_$tag.json 2> gpurun_out/r5b/c5s_$tag.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r5b/c5s_$tag.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$tag", round(d["ms_per_step"]/1e3,2), "s", c["gcg_iterations"], "its", c["nev_converged"], "pairs", c["cg_iterations"], "cg", c.get("amg_levels"), flush=True)
except Exception as e:
    print("$tag failed / timed out", e, flush=True)
P
}
run plain GCGE_X=0 --amg 0
run amg3_8_24 GCGE_X=0 --amg 3 --amg-smooth 8,24
run amg3_8_24_slots GCGE_AMG_NO_FUSIONS=1 --amg 3 --amg-smooth 8,24
run amg3_5_4 GCGE_X=0 --amg 3 --amg-smooth 5,4
run amg2_8_24 GCGE_X=0 --amg 2 --amg-smooth 8,24
