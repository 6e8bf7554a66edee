// Internal helpers shared by the HIP translation units of libgcge_hip.so.
#ifndef GCGE_HIP_INTERNAL_H
#define GCGE_HIP_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define GCGE_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "gcge_hip: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, \
              hipGetErrorString(e_));                                                 \
      abort();                                                                        \
    }                                                                                 \
  } while (0)

#endif
