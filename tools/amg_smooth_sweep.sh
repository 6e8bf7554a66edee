# C2 with BlockAMG: smoothing steps (finest level, coarser levels) and merged column passes on / off — one line per run
set -e
mkdir -p gpurun_out/r5b
run() {  # tag, env, args...
  tag=$1; shift; envs=$1; shift
  env $envs python bench.py --steps 1 --warmup 1 --no-extra --no-cpu "$@" > gpurun_out/r5b/sweep_$tag.json 2> gpurun_out/r5b/sweep_$tag.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5b/sweep_$tag.json").read().strip().splitlines()[-1])
c=d["config"]
print("$tag", "value", round(d["value"],3), "s/solve", round(d["ms_per_step"]/1e3,3), "its", c["gcg_iterations"], "conv", c["nev_converged"], "cg", c["cg_iterations"], "linsol", round(c["phase_seconds"]["linsol"],3), flush=True)
P
}
run merge0_3,4 GCGE_PASS_MERGE=0 --amg-smooth 3,4
run merge256_3,4 GCGE_PASS_MERGE=256 --amg-smooth 3,4
for sm in 2,4 2,3 3,3 3,2 2,2 4,4; do run m256_$sm GCGE_PASS_MERGE=256 --amg-smooth $sm; done
