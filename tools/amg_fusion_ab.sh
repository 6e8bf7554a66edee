# C2 with BlockAMG: the V-cycle's fused residual / correction on and off — one line per run
set -e
mkdir -p gpurun_out/r5b
run() {  # tag, env, args...
  tag=$1; shift; envs=$1; shift
  env $envs python bench.py --steps 2 --warmup 1 --no-extra --no-cpu "$@" > gpurun_out/r5b/ab_$tag.json 2> gpurun_out/r5b/ab_$tag.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5b/ab_$tag.json").read().strip().splitlines()[-1])
c=d["config"]
print("$tag", "value", round(d["value"],3), "s/solve", round(d["ms_per_step"]/1e3,3), "its", c["gcg_iterations"], "conv", c["nev_converged"], "cg", c["cg_iterations"], "linsol", round(c["phase_seconds"]["linsol"],3), flush=True)
P
}
run slots GCGE_AMG_NO_FUSIONS=1
run fused GCGE_AMG_NO_FUSIONS_OFF=1
