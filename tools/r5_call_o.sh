#!/bin/bash
# round 5, call O: the collective's workgroup sweep with one region per process; the default bench line with the stamp of the final sources
export TMPDIR=/tmp
o=gpurun_out/r5q
mkdir -p $o
timeout -k 10 900 python tools/p2p_sweep.py > $o/p2p_sweep.jsonl 2> $o/p2p_sweep.err
grep -v "Gloo\|Feature\|amdgpu\|socket\|^$" $o/p2p_sweep.err | tail -5
cat $o/p2p_sweep.jsonl
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); print('C3', d['value'], d['ms_per_step'], d['roofline'])
"
