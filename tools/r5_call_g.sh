#!/bin/bash
# round 5, call G: the whole GPU suite + the bench line on the build with the fused sampler / transform, the
# K hint of the layout builder and four entries in flight per lane group at small K
export TMPDIR=/tmp
o=gpurun_out/r5i
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); a=d['also']
print('C3', d['ms_per_step'], d['kernel_ms'], 'vi', d['vi_step_ms'])
for k in sorted(a):
    if any(s in k for s in ('shard125k_vi', 'shard125k_ms', 'c1_gpu', 'ref_harness', 'c2_ms', 'minibatch_ms')): print('  ', k, a[k])
"
python tools/fixed_cost_probe.py 2 > $o/fixed.jsonl 2> $o/fixed.err || { tail -5 $o/fixed.err; exit 1; }
