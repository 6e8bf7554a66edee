mkdir -p gpurun_out/r5b
run() { tag=$1; shift; envs=$1; shift
  env $envs timeout -k 10 200 python bench.py --config c5 --size 96 --atoms 350,2.0,5.0 --steps 1 --warmup 0 --no-cpu --no-extra --amg 4 --amg-smooth 8,24 > gpurun_out/r5b/c5h_$tag.json 2> gpurun_out/r5b/c5h_$tag.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r5b/c5h_$tag.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$tag", round(d["ms_per_step"]/1e3,2), "s", c["gcg_iterations"], "its", c["nev_converged"], "pairs", c["cg_iterations"], "cg", flush=True)
except Exception as e:
    print("$tag failed", e, flush=True)
P
}
run device GCGE_X=0
run host_scalars GCGE_CG_HOST_SCALARS=1
run stored_host GCGE_CG_STORED_HOST=1
