#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/pmc.sh into profiles/pmc_traffic.json.

    tools/pmc.sh gpurun_out/pmc --workload c3     # separate --pmc passes, no trace domains
    python tools/pmc_traffic.py gpurun_out/pmc c3 profiles/pmc_traffic.json

Per kernel and launch: FETCH_SIZE / WRITE_SIZE (KB, as reported), the L2 hit rate, and
the HBM-side traffic bench.py quotes as roofline.traffic:
    traffic = 2 * FETCH_SIZE + WRITE_SIZE      [bytes]
following MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half
the bytes of 16-B-per-lane reads (TCC_EA0_RDREQ x 64 B for 128-B requests) -- every
gather and stream of these kernels is a 16-B-per-lane load -- and WRITE_SIZE is exact
for 16-B stores and float atomics.  Infinity-Cache hits are counted (it is traffic
beyond the XCD's L2, not necessarily HBM).  The file is stamped with a hash of the
kernel sources; bench.py ignores it when the sources have changed since.
"""
import csv
import collections
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spmf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


SHORT = {"row_pass_kernel": "row_pass", "col_pass_kernel": "col_pass", "col_pass_wide_kernel": "col_pass", "sigdot3_kernel": "dense_expdot",
         "expdot_kernel": "dense_expdot", "expdot3_kernel": "dense_expdot", "finish_kernel": "finish",
         "prep_kernel": "prep", "begin_kernel": "prep", "end_kernel": "finish"}


def main():
    out_dir, workload, dest = sys.argv[1], sys.argv[2], sys.argv[3]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for pat, short in SHORT.items():
                if pat in k:
                    agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    doc = {}
    if os.path.exists(dest):
        try:
            doc = json.load(open(dest))
        except Exception:
            doc = {}
    if doc.get("_kernels_sha") != kernels_sha():
        doc = {}                                   # other workloads' entries are stale too
    doc["_kernels_sha"] = kernels_sha()
    doc["_formula"] = "traffic_bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (MI355X_MICROARCH.md, HBM)"
    entry = {}
    for k, c in agg.items():
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        e = {"launches_sampled": max(len(v) for v in c.values())}
        if "FETCH_SIZE" in mean:
            e["fetch_size_kb"] = mean["FETCH_SIZE"]
        if "WRITE_SIZE" in mean:
            e["write_size_kb"] = mean["WRITE_SIZE"]
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            e["traffic_bytes"] = 2 * mean["FETCH_SIZE"] * 1024 + mean["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
            e["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
        for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
            if n in mean:
                e[n] = mean[n]
        entry[k] = e
    doc[workload] = entry
    json.dump(doc, open(dest, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
