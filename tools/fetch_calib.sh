#!/bin/bash
# FETCH_SIZE against known byte counts (tools/fetch_calib.hip).  usage (on the GPU box): tools/fetch_calib.sh <outdir>
export TMPDIR=/tmp
out=${1:-gpurun_out/fetch_calib}
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p -- tools/bin/fetch_calib 4 > $out/run.txt 2> $out/run.err || tail -5 $out/run.err
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
known = {"calib_stream4": 4 * 2**30, "calib_stream": 4 * 2**30, "calib_gather<8>": 4 * 2**30 + (4 * 2**30 // 128) * 4,
         "calib_gather<16>": 4 * 2**30 + (4 * 2**30 // 256) * 4}
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for k, b in known.items():
            if name == k or name.startswith(k + "<"):
                kb = float(r["Counter_Value"])
                print(f"{k:18s} FETCH_SIZE {kb:14.0f} KB = {kb * 1024 / b:6.3f} x the known {b / 1e9:.3f} GB"
                      f"  ->  bytes = {b / (kb * 1024):.3f} x FETCH_SIZE")
PY
