#!/usr/bin/env python3
"""profiles/r05_pmc_sparse_passes.txt: the counter passes of tools/pmc_r05.sh (any number of output directories)
merged per kernel, with the derived figures DESIGN.md section 4 quotes.
usage: pmc_r05_report.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for out in sys.argv[1:]:
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void spmf::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void spmf::", "")
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
for k in ("row_pass_kernel", "col_pass_kernel"):
    e = {c: sum(v) / len(v) for c, v in agg[k].items() if sum(v) / len(v) > 0 or c.startswith("TCC")}
    big = [d for d in dur[k] if d > 0.5 * max(dur[k])]            # the C3-sized launches
    d = sum(big) / len(big)
    print(f"== {k}: {len(big)} launches on C3 (1e8 stored entries, K = 32), mean {d * 1e3:.4f} ms under the profiler")
    for c in sorted(agg[k]):
        v = [x for x in agg[k][c]]
        top = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        m = sum(top) / len(top)
        e[c] = m
        print(f"   {c:44s} {m:.6g}")
    g = e.get("TCP_GATE_EN1_sum")
    if g and e.get("TCP_TCC_READ_REQ_sum"):
        req, lat = e["TCP_TCC_READ_REQ_sum"], e["TCP_TCC_READ_REQ_LATENCY_sum"]
        print("   -- derived, vector cache (TCP) side, 256 CUs:")
        print(f"      L1-miss read requests per clock and CU      {req / g:.3f}   (128-B requests: {128 * req / g:.1f} B/clk/CU; "
              f"the guide's L2 peak 34.5 TB/s is 64 B/clk/CU)")
        print(f"      mean latency of a request                   {lat / req:.0f} clk")
        print(f"      requests in flight per CU (latency sum / busy cycles)  {lat / g:.1f}")
        print(f"      vector cache stalled on a pending line      {100 * e['TCP_PENDING_STALL_CYCLES_sum'] / g:.1f} % of its busy cycles")
        print(f"      tag accesses per request                    {e['TCP_TOTAL_CACHE_ACCESSES_sum'] / req:.2f}")
    if e.get("TCC_CYCLE_sum"):
        cyc = e["TCC_CYCLE_sum"]
        print("   -- derived, L2 (TCC) side, 128 channels:")
        print(f"      L2 busy                                     {100 * e['TCC_BUSY_sum'] / cyc:.1f} % of channel cycles")
        print(f"      requests per channel cycle                  {e['TCC_REQ_sum'] / cyc:.3f}   ({cyc / e['TCC_REQ_sum']:.2f} channel cycles per request)")
        print(f"      hit rate                                    {100 * e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum']):.1f} %")
        print(f"      fabric read requests (TCC_EA0_RDREQ)        {e['TCC_EA0_RDREQ_sum']:.4g}  (x 128 B = {e['TCC_EA0_RDREQ_sum'] * 128 / 1e9:.2f} GB)")
        print(f"      tag stalls                                  {100 * e['TCC_TAG_STALL_sum'] / cyc:.2f} % of channel cycles")
    if e.get("SQ_WAVE_CYCLES"):
        print("   -- derived, SQ:")
        print(f"      waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)  {100 * e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']:.1f} %   issue stalls {100 * e['SQ_WAIT_INST_ANY'] / e['SQ_WAVE_CYCLES']:.1f} %")
        print(f"      clock (GRBM_GUI_ACTIVE / 8 / duration)       {e['GRBM_GUI_ACTIVE'] / 8 / d / 1e9:.2f} GHz")
    print()
