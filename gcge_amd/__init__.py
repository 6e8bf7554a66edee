"""gcge_amd — MI355X-native back-end for the GCG eigensolver hot path.

Python is plumbing only (ctypes over the C ABI declared in include/*.h): device
kernels are HIP (gcge_amd/csrc/hip), host logic is C (gcge_amd/csrc/host).
"""
from .lib import (CSR, RunResult, Timing, host_lib, hip_lib, build_libs,
                  make_problem, load_petsc_binary, load_matrix_market, HipBackend, run_gcg)

__all__ = ["CSR", "RunResult", "Timing", "host_lib", "hip_lib", "build_libs",
           "make_problem", "load_petsc_binary", "load_matrix_market", "HipBackend", "run_gcg"]
