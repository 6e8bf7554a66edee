# config 5 with BlockAMG on 2 row slabs sharing one GPU, deeper hierarchies: where does it stop converging?  (bash tools/c5_slab_amg_probe2.sh SIZE ATOMS)
mkdir -p gpurun_out/r5b
SZ=${1:-64}; AT=${2:-100,2.0,5.0}
run() { tag=$1; shift; envs=$1; shift
  env $envs timeout -k 10 70 python bench.py --config c5 --gpus 2 --rehearse --size $SZ --atoms $AT --steps 1 --no-cpu --no-extra "$@" > gpurun_out/r5b/c5t_$tag.json 2> gpurun_out/r5b/c5t_$tag.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r5b/c5t_$tag.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$tag", round(d["ms_per_step"]/1e3,2), "s", c["gcg_iterations"], "its", c["nev_converged"], "pairs", c["cg_iterations"], "cg", c.get("amg_levels"), flush=True)
except Exception as e:
    print("$tag failed / timed out", flush=True)
P
}
run amg3 GCGE_MG_TRACE=1 --amg 3 --amg-smooth 8,24
run amg3_host_smoother GCGE_AMG_HOST_SMOOTHER=1 --amg 3 --amg-smooth 8,24
run amg3_host_scalars GCGE_CG_HOST_SCALARS=1 --amg 3 --amg-smooth 8,24
