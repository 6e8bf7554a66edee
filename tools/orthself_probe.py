"""Time the three slot calls the reference's column-wise OrthSelf makes per column (src/ops_orth.c:45-118) on a 64-column
panel of a 256^3-row block: QtAP('S','N') of the panel against its first column, the scaling of that column, the rank-1
update of the others — against the bytes they must move.    python tools/orthself_probe.py [N] [b]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = HipBackend(); ops = hip.ops
A, _ = make_problem("lap3d", N)
mA = hip.matrix(A)
n = A.nrows
hip.set_random_mode(1, 3)
V = ops.mv_create(256, mA); ops.set_random(V, 0, 256)
ws = ops.mv_create(b, mA)
def timed(fn, reps=5):
    fn(); hip.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    hip.sync()
    return (time.perf_counter() - t0) / reps
tot = 0.0
for k0, w in ((128, b), (128 + b // 2, b // 2), (128 + b - 8, 8), (128 + b - 2, 2)):
    e = k0 + w
    coef = np.full(w - 1 if w > 1 else 1, 0.01)
    t_q = timed(lambda: ops.qtap("S", "N", V, None, V, (k0, k0), (e, k0 + 1), ws, ld=w))
    t_s = timed(lambda: ops.axpby(0.0, None, 1.0000001, V, (k0, k0), (k0 + 1, k0 + 1)))
    t_u = timed(lambda: ops.lincomb(V, V, (k0, k0 + 1), (k0 + 1, e), coef, 1, beta=np.ones(1), incb=0)) if w > 1 else 0.0
    gb = 8.0 * n * 1e-9
    print("panel of %3d columns: QtAP %.3f ms (%.0f GB/s on %d columns read)  scale %.3f ms  rank-1 update %.3f ms (%.0f GB/s on %d read + written)"
          % (w, 1e3 * t_q, w * gb / t_q, w, 1e3 * t_s, 1e3 * t_u, (2 * w - 1) * gb / max(t_u, 1e-9), 2 * w - 1), flush=True)
