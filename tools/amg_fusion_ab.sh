# C2 with BlockAMG: A/B of the round-5 switches, one line per run:  bash tools/amg_fusion_ab.sh TAG=ENV[,ENV...] ...
set -e
mkdir -p gpurun_out/r5b
run() {  # tag, env, args...
  tag=$1; shift; envs=$1; shift
  env $envs python bench.py --steps 2 --warmup 1 --no-extra --no-cpu "$@" > gpurun_out/r5b/ab_$tag.json 2> gpurun_out/r5b/ab_$tag.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5b/ab_$tag.json").read().strip().splitlines()[-1])
c=d["config"]
print("$tag", "value", round(d["value"],3), "s/solve", round(d["ms_per_step"]/1e3,3), "its", c["gcg_iterations"], "conv", c["nev_converged"], "cg", c["cg_iterations"], "linsol", round(c["phase_seconds"]["linsol"],3), "roofline", round(d["roofline"]["frac"],3), flush=True)
P
}
for spec in "$@"; do
  tag=${spec%%=*}; envs=${spec#*=}
  run $tag "$(echo $envs | tr ',' ' ')"
done
