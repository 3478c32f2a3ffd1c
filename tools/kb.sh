#!/bin/bash
# usage: tools/kb.sh [bench args]   -> prints ms/step and per-kernel ms
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/kb.json 2> gpurun_out/kb.err || { tail -20 gpurun_out/kb.err; exit 1; }
python -c '
import json
d=json.load(open("gpurun_out/kb.json"))
print("ms/step", round(d["ms_per_step"],4), d["kernel_ms"], "frac_step", round(d["frac_hbm_roofline_step"],4))
'
