"""Row-partitioned (one process per GPU) execution of the GCG hot path — see DESIGN.md §multi-GPU."""


def install(hip, dist, rank, world):
    raise NotImplementedError("multi-GPU path: see gcge_amd/csrc/hip/dist.hip (in progress)")


def lap3d_slab(hip, N, planes, rank, world):
    raise NotImplementedError
