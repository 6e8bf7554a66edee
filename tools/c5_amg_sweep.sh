# config 5 on one GPU with BlockAMG: smoothing steps (finest level, coarser levels) — one line per run
mkdir -p gpurun_out/r5b
for sm in "$@"; do
  python bench.py --config c5 --steps 1 --warmup 0 --no-cpu --no-extra --amg 5 --amg-smooth $sm > gpurun_out/r5b/c5_sweep_$sm.json 2> gpurun_out/r5b/c5_sweep_$sm.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5b/c5_sweep_$sm.json").read().strip().splitlines()[-1])
c=d["config"]
print("c5 amg 5 smooth $sm:", round(d["ms_per_step"]/1e3,2), "s", c["gcg_iterations"], "its", c["nev_converged"], "pairs", c["cg_iterations"], "cg its, linsol", round(c["phase_seconds"]["linsol"],1), flush=True)
P
done
