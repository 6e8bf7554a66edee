"""K7 timing probe: gcge_hip_symeig on the Rayleigh-Ritz-shaped matrix of tests/test_hip_parity.py, repeated.
    python tools/eig_probe.py [n] [repeats]        (GCGE_EIG_TIMING=1 prints the phase times of every call on stderr)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from gcge_amd import HipBackend   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 656
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hip = HipBackend(); g = hip.g
g.gcge_hip_symeig.argtypes = [C.c_char, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
rng = np.random.default_rng(7)
A = np.diag(np.sort(rng.random(n)) * 3.0)
b = min(256, n)
Bd = rng.standard_normal((n, b)) * 0.1
A[:, n - b:] += Bd; A[n - b:, :] += Bd.T
A = np.asfortranarray((A + A.T) * 0.5)
w = np.zeros(n); z = np.zeros((n, n), order="F")
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    assert g.gcge_hip_symeig(b"U", n, A.ctypes.data, n, w.ctypes.data, z.ctypes.data, n) == 0
    ts.append(1e3 * (time.perf_counter() - t0))
ref = np.linalg.eigvalsh(A)
print("n=%d: eigenvalue error %.2e, residual %.2e, orthogonality %.2e" % (
    n, np.abs(w - ref).max(), np.abs(A @ z - z * w).max(), np.abs(z.T @ z - np.eye(n)).max()))
print("n=%d: calls (ms) %s   median %.2f  min %.2f" % (n, " ".join("%.1f" % t for t in ts), float(np.median(ts)), min(ts)))
