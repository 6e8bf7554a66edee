#!/usr/bin/env python3
"""Fabric bytes of the K1 product of BASELINE config 5's matrix, per kernel of the product (plane sweep, dense blocks, listed rows),
from rocprofv3 PMC passes over tools/k1_c5_probe.py; merged into profiles/pmc_traffic.json as entry "c5", keyed by a content hash
of the kernel sources (bench.py quotes it as roofline_k1_c5.traffic only while the hash matches).

    python3 tools/pmc_traffic_c5.py gpurun_out/r5/pmc_c5          (on the GPU box; then copy the JSON to profiles/)

Counters and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: TCC_EA0_RDREQ x 128 B for 16-byte-per-lane
loads (FETCH_SIZE tallies them at 64 B), TCC_EA0_WRREQ x 64 B; reads and writes in separate passes, no tracing options beside --pmc."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["spmm_dense.hip", "spmm_pad8.hip", "spmm_star.hip"]


def source_hash():
    h = hashlib.sha256()
    for f in sorted(FILES):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "gcge_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()


def main():
    out = os.path.abspath(sys.argv[1])
    G, K, m, nprod = 171, 2000, 64, 4
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    groups = ["TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum", "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"]
    for i, grp in enumerate(groups, 1):
        cmd = ["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", os.path.join(out, "p%d" % i), "--",
               "python3", os.path.join(ROOT, "tools", "k1_c5_probe.py"), str(G), str(K), str(m), str(nprod)]
        with open(os.path.join(out, "log%d.txt" % i), "w") as lf:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=lf, stderr=subprocess.STDOUT, check=True)
    acc = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if not any(t in k for t in ("spmm_star", "dense", "pad8", "spmm_tile")):
                continue
            short = k.split("(")[0].replace("void ", "").replace("gcge::", "").strip()
            acc[(short, r["Counter_Name"])] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
                launches[(short, r["Counter_Name"])] += 1
    kernels = {}
    for (kn, cn), v in acc.items():
        kernels.setdefault(kn, {})[cn] = v / nprod            # per PRODUCT (a kernel may be launched several times per product)
    total = 0.0
    for kn, c in kernels.items():
        rd, wr = c.get("TCC_EA0_RDREQ_sum"), c.get("TCC_EA0_WRREQ_sum")
        c["launches_per_product"] = launches.get((kn, "TCC_EA0_RDREQ_sum"), 0) / nprod
        if rd is not None and wr is not None:
            c["fabric_bytes_per_block_operation"] = rd * 128.0 + wr * 64.0
            total += c["fabric_bytes_per_block_operation"]
    entry = {"source_files": FILES, "source_sha256": source_hash(), "shape": {"G": G, "atoms": "%d,2.0,5.0" % K, "m": m},
             "bytes_rule": "per kernel of the product: TCC_EA0_RDREQ x 128 B (wide coalesced reads are 128-byte requests tallied at 64: MI355X_MICROARCH.md) "
                           "+ TCC_EA0_WRREQ x 64 B; the dense blocks' gathers are 128-byte requests as well",
             "profile": "tools/pmc_traffic_c5.py (rocprofv3 --pmc over tools/k1_c5_probe.py, %d products)" % nprod,
             "kernels": kernels, "product_fabric_bytes": total}
    with open(os.path.join(out, "pmc_traffic_c5.json"), "w") as f:
        json.dump(entry, f, indent=1, sort_keys=True)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        allt = json.load(open(path))
        allt["c5"] = entry
        with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
            json.dump(allt, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
    for kn in sorted(kernels):
        print(kn, kernels[kn])
    print("product:", total)


if __name__ == "__main__":
    main()
