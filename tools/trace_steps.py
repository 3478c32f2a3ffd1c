#!/usr/bin/env python3
"""Per-step kernel durations and the gaps between them from a rocprofv3 --kernel-trace csv.
usage: trace_steps.py <dir with *kernel_trace.csv> [first_kernel_substr] [last_kernel_substr]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "begin_kernel"
last = sys.argv[3] if len(sys.argv) > 3 else "end_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = r.get("Grid_Size_X") or r.get("Grid_Size")
    short = n.split("(")[0].replace("void spmf::", "")
    if first in n:
        cur = [(short, s, e, g)]
    elif cur is not None:
        cur.append((short, s, e, g))
        if last in n:
            steps.append(cur)
            cur = None
groups = collections.defaultdict(list)
for st in steps:
    groups[tuple((k[0][:34], k[3]) for k in st)].append(st)
for key, sts in groups.items():
    if len(sts) < 20:
        continue
    sts = sts[10:]
    n = len(sts)
    print("---", n, "steps")
    for i in range(len(key)):
        dur = sum(st[i][2] - st[i][1] for st in sts) / n
        gap = sum(st[i][1] - st[i - 1][2] for st in sts) / n if i else 0
        print(f"  {key[i][0]:36s} grid {key[i][1]:>9s} dur {dur / 1e3:8.2f} us  gap_before {gap / 1e3:6.2f} us")
    tot = sum(st[-1][2] - st[0][1] for st in sts) / n
    per = sorted(p for p in ((sts[j + 1][0][1] - sts[j][0][1]) for j in range(n - 1)) if p < 5e6)
    print(f"  first->last span {tot / 1e3:.2f} us ; step period median {per[len(per) // 2] / 1e3:.2f} us")
