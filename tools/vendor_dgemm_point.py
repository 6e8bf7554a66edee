"""Datapoint, not part of the product: what the vendor GEMM (torch.matmul -> rocBLAS / hipBLASLt) reaches on the
shapes of the panel update (Y[n x m] = X[n x k] C[k x m]) and the Gram product (G[k x m] = Q[n x k]^T P[n x m]),
row-major f64, n = 2^24 — the yardstick for the hand-written FP64-MFMA kernels (tools/dense_bench.hip)."""
import time
import torch
n = 1 << 24
dev = torch.device("cuda")
for k, m in ((256, 64), (256, 128), (192, 64), (64, 64)):
    X = torch.rand(n, k, dtype=torch.float64, device=dev) - 0.5
    Cm = torch.rand(k, m, dtype=torch.float64, device=dev) - 0.5
    P = torch.rand(n, m, dtype=torch.float64, device=dev) - 0.5
    for name, fn in (("panel update  X C ", lambda: torch.matmul(X, Cm)), ("Gram          X^T P", lambda: torch.matmul(X.t(), P))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("vendor %s k=%3d m=%3d  %8.3f ms  %6.1f TF" % (name, k, m, ms, 2.0 * n * k * m / ms * 1e-9), flush=True)
    del X, Cm, P
