for lv in 3 4 6; do
  python bench.py --config c5 --steps 1 --warmup 0 --no-cpu --no-extra --amg $lv --amg-smooth 8,24 > gpurun_out/r5b/c5_lv$lv.json 2> gpurun_out/r5b/c5_lv$lv.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5b/c5_lv$lv.json").read().strip().splitlines()[-1]); c=d["config"]
print("c5 amg $lv smooth 8,24:", round(d["ms_per_step"]/1e3,2), "s", c["gcg_iterations"], "its", c["nev_converged"], "pairs", c["cg_iterations"], "cg its, linsol", round(c["phase_seconds"]["linsol"],1), "setup", round(c["amg_setup_seconds"],1), flush=True)
P
done
