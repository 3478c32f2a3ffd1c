#!/usr/bin/env python3
"""Per-kernel means of the counter passes of tools/pmc_r05.sh, with the derived request rates.

    python3 tools/pmc_r05_summary.py gpurun_out/pmc5 [json out]

Prints, for row_pass / col_pass: every counter (mean per launch), the kernel's average duration
from the kernel traces of the same passes, and
    L1-miss read requests per clock and CU = TCP_TCC_READ_REQ / (duration * clock * 256)
against the vector cache's 64 B/clk/CU (MI355X_MICROARCH.md, L2: 34.5 TB/s), taking a request
as 64 B when TCC_EA-side counters say so (the guide's gfx950 correction) and reporting both.
"""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void spmf::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void spmf::", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
doc = {}
for k in sorted(agg):
    if not any(s in k for s in ("row_pass", "col_pass", "finish", "prep", "begin")):
        continue
    d = sum(dur[k]) / max(1, len(dur[k]))
    print(f"== {k}   launches {len(dur[k])}   mean duration {d * 1e3:.4f} ms (under the profiler)")
    e = {"duration_ms_profiled": d * 1e3}
    for c, v in sorted(agg[k].items()):
        m = sum(v) / len(v)
        e[c] = m
        print(f"   {c:44s} n={len(v):3d} mean={m:.6g}")
    doc[k] = e
if len(sys.argv) > 2:
    json.dump(doc, open(sys.argv[2], "w"), indent=1, sort_keys=True)
