"""(script, not collected by pytest)  tools/run_case.py plus the compiled reference (oracle/_ref) on the same input:
adds converged count, iteration count, wall time of the reference CPU path and the largest relative difference of the
Ritz values to the JSON line.  Lives under tests/ because only tests may load anything from oracle/.
    python tests/case_vs_reference.py --kind lap3d --size 50 --nev 20 --block 20 --rng 0      # BASELINE config 1"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, _p)
import run_case  # noqa: E402


def with_reference(out, a, A, B, ev, k):
    import pyoracle as po
    if po.ref_lib() is None:
        out["ref"] = "oracle/_ref not built"
        return
    rv, conv, it, sec = po.ref_gcg(A, B, a.nev, nev_max=a.nevmax, block=a.block)
    kk = min(conv, k)
    out["ref"] = {"nev_converged": conv, "gcg_iterations": it, "seconds": sec,
                  "max_rel_diff_ritz_values": float(np.max(np.abs(ev[:kk] - rv[:kk]) / np.abs(rv[:kk])))}


if __name__ == "__main__":
    run_case.main(post=with_reference)
