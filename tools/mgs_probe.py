"""Rate of the one-sweep Gram-Schmidt step (vec_kernels.hip: mgs_step_kernel) on a 256^3-row panel of a 256-column block, per
width w of the columns behind the step's own, and its Gram column against the separate kernel (bitwise).
    python tools/mgs_probe.py [N]
Finding (profiles/r03_dropin/39_mgs_probe.log): the time is a step function of the 128-byte lines a row's segment touches —
about 1.0 ms per line (read + written) for 256^3 rows, i.e. 4.3 TB/s of whole-line traffic whatever the unroll depth or the
lane layout; the columns of a panel share their lines with the rest of the 2 KB row."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gcge_amd import HipBackend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, ld = N ** 3, 256
hip = HipBackend(); g = hip.g
g.gcge_hip_mgs_step.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
g.gcge_hip_panel_dot1.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
g.gcge_hip_stream.restype = C.c_void_p
st = g.gcge_hip_stream()
torch.manual_seed(1)
V = torch.rand((n, ld), dtype=torch.float64, device="cuda")
cvec = torch.full((64,), -1e-3, dtype=torch.float64, device="cuda")
dots = torch.zeros(64, dtype=torch.float64, device="cuda"); dots2 = torch.zeros(64, dtype=torch.float64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for w in (63, 56, 48, 40, 33, 32, 24, 16, 9, 8, 4, 2, 1):
    k0 = 100
    base = V.data_ptr() + 8 * k0
    def step():
        g.gcge_hip_mgs_step(n, base, ld, 1.0000001, cvec.data_ptr(), w, dots.data_ptr(), st)
    step(); hip.sync(); torch.cuda.synchronize()
    reps = 5
    import time
    t0 = time.perf_counter()
    for _ in range(reps): step()
    hip.sync()
    dt = (time.perf_counter() - t0) / reps
    # the separate kernel on the updated panel: column k0+1 against the w columns behind the step's own
    g.gcge_hip_panel_dot1(n, base + 8, ld, w, base + 8, ld, dots2.data_ptr(), st)
    hip.sync()
    same = bool(torch.equal(dots[:w], dots2[:w]))
    gb = (2 * w + 2) * 8.0 * n * 1e-9
    print("w=%2d: %.3f ms  %.0f GB/s on %d columns read + written   Gram column bitwise equal to panel_dot1: %s" % (w, 1e3 * dt, gb / dt, 2 * w + 2, same), flush=True)
