#!/bin/bash
# round 5, call K: sanity after the last Python-side changes (make_reducer, bench fields): p2p / two-rank / launch tests, the default bench line
export TMPDIR=/tmp
o=gpurun_out/r5m
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_two_rank.py tests/test_gpu_bench_launch.py tests/test_gpu_driver.py -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); a=d['also']
print('C3', d['value'], d['ms_per_step'], d['roofline'])
for k in sorted(a):
    if 'c4_' in k: print('  ', k, a[k])
"
