import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gcge_amd import HipBackend, make_problem
hip = HipBackend(); ops = hip.ops
A, _ = make_problem("lap3d", 256); mA = hip.matrix(A)
hip.set_random_mode(1, 7)
x = ops.mv_create(128, mA); y = ops.mv_create(128, mA); ops.set_random(x, 0, 128); ops.set_random(y, 0, 128)
for m in (64, 128):
    for (a, b, tag) in ((1.0, 0.0, "copy"), (2.0, 0.5, "axpby"), (0.0, 3.0, "scale")):
        xx = None if a == 0.0 else x
        ops.axpby(a, xx, b, y, (0, 0), (m, m)); hip.sync()
        t = time.perf_counter()
        for _ in range(10): ops.axpby(a, xx, b, y, (0, 0), (m, m))
        hip.sync(); ms = (time.perf_counter() - t) / 10 * 1e3
        streams = 2 if tag != "axpby" else 3
        print("%s m=%d: %.3f ms  %.0f GB/s" % (tag, m, ms, streams * 8.0 * A.nrows * m / ms * 1e-6), flush=True)
