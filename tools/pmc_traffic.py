#!/usr/bin/env python3
"""Fabric (HBM-side) bytes per launch of the SpMM / CG-pass kernels at the bench shape, from rocprofv3 PMC passes,
written as a JSON that bench.py quotes as `roofline.traffic` — keyed by kernel and by a content hash of the HIP
sources, so a number measured on other kernels is never quoted (bench.py refuses a stale file).

    python3 tools/pmc_traffic.py gpurun_out/r2/pmc [N] [m]         (on the GPU box; then copy the JSON to profiles/)

Counters and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: TCC_EA0_RDREQ counts 128-B requests
for the 16-byte-per-lane loads these kernels issue (FETCH_SIZE tallies them at 64 B: half the bytes), TCC_EA0_WRREQ
x 64 B is exact for 16-byte-per-lane stores; reads and writes in separate passes (4 TCC slots per pass), no tracing
options next to --pmc."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """sha256 over the HIP sources of libgcge_hip.so (sorted by name): what `traffic` is valid for."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gcge_amd", "csrc", "hip")
    # the sources that define the profiled kernels (SpMM / CG passes): an edit to the eigensolver, to another matrix form or
    # to the CG's host loop does not change what a launch of these kernels moves
    for f in sorted(os.listdir(d)):
        if f in ("spmm_pattern.hip", "spmm_ring.hip"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def main():
    out = os.path.abspath(sys.argv[1])
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    groups = ["TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum",
              "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"]
    for i, grp in enumerate(groups, 1):
        cmd = ["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", os.path.join(out, "p%d" % i), "--",
               "python3", os.path.join(ROOT, "tools", "cg_pass_probe.py"), str(N), str(m)]
        with open(os.path.join(out, "log%d.txt" % i), "w") as lf:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=lf, stderr=subprocess.STDOUT, check=True)
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            mm = re.search(r"(spmm_\w+?)(?:_kernel)?<([^>]*)>", k)
            if not mm:
                continue
            acc[(mm.group(1) + "<" + mm.group(2).replace(" ", "") + ">", r["Counter_Name"])].append(float(r["Counter_Value"]))
    kernels = {}
    for (kn, cn), v in acc.items():
        kernels.setdefault(kn, {})[cn] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    npass = (m + 15) // 16
    for kn, c in kernels.items():
        rd = c.get("TCC_EA0_RDREQ_sum", {}).get("mean_per_launch")
        wr = c.get("TCC_EA0_WRREQ_sum", {}).get("mean_per_launch")
        if rd is not None and wr is not None:
            c["fabric_bytes_per_kernel_launch"] = rd * 128.0 + wr * 64.0
            c["fabric_bytes_per_block_operation"] = npass * (rd * 128.0 + wr * 64.0)   # one m-column op = npass launches of 16 columns
    res = {"source_sha256": source_hash(), "shape": {"problem": "lap3d", "N": N, "m": m, "passes_per_operation": npass},
           "bytes_rule": "TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B (MI355X_MICROARCH.md, HBM: 16-byte-per-lane accesses)",
           "probe": "tools/cg_pass_probe.py %d %d" % (N, m), "kernels": kernels}
    with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for kn in sorted(kernels):
        print(kn, kernels[kn].get("fabric_bytes_per_block_operation"))
    print(open(os.path.join(out, "log1.txt")).read()[-400:])


if __name__ == "__main__":
    main()
