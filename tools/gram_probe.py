"""Time gcge_hip_gram (gram_mfma.hip) at the shapes a config 2 solve uses, against torch for the values.
    python tools/gram_probe.py [N]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcge_amd import HipBackend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = N ** 3
hip = HipBackend(); g = hip.g
g.gcge_hip_gram.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
g.gcge_hip_stream.restype = C.c_void_p
st = g.gcge_hip_stream()
torch.manual_seed(3)
Q = torch.rand((n, 256), dtype=torch.float64, device="cuda") - 0.5
P = torch.rand((n, 64), dtype=torch.float64, device="cuda") - 0.5
for k, m in ((192, 64), (256, 64), (128, 64), (64, 64), (384, 64)):
    if k > 256:
        continue
    G = torch.zeros((k, m), dtype=torch.float64, device="cuda")
    call = lambda: g.gcge_hip_gram(n, Q.data_ptr(), 256, k, P.data_ptr(), 64, m, G.data_ptr(), st)
    call(); hip.sync()
    t0 = time.perf_counter()
    for _ in range(5): call()
    hip.sync()
    dt = (time.perf_counter() - t0) / 5
    rows = min(n, 1 << 20)
    ref = Q[:rows, :k].T @ P[:rows, :m]
    g.gcge_hip_gram(rows, Q.data_ptr(), 256, k, P.data_ptr(), 64, m, G.data_ptr(), st); hip.sync()
    err = float((G - ref).abs().max() / ref.abs().max())
    print("k=%3d m=%2d: %.3f ms  %.1f TF  %.2f TB/s on 8 n (k + m)   max rel diff vs torch on 2^20 rows %.1e" % (
        k, m, 1e3 * dt, 2.0 * n * k * m / dt * 1e-12, 8.0 * n * (k + m) / dt * 1e-12, err), flush=True)
