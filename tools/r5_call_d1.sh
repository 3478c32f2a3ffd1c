#!/bin/bash
# round 5, call D1: the whole GPU suite on the row pass with the dynamic tail, the fixed-cost probe with it on / off, the bench line
export TMPDIR=/tmp
o=gpurun_out/r5e
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
python tools/fixed_cost_probe.py 3 > $o/fixed_dyn.jsonl 2> $o/fixed_dyn.err || { tail -5 $o/fixed_dyn.err; exit 1; }
SPMF_ROW_DYNAMIC=0 python tools/fixed_cost_probe.py 3 > $o/fixed_static.jsonl 2> $o/fixed_static.err || { tail -5 $o/fixed_static.err; exit 1; }
for f in $o/fixed_dyn.jsonl $o/fixed_static.jsonl; do echo $f; python3 -c "
import json
for l in open('$f'):
    d=json.loads(l); r=d['runs'][-1]; print('  ', d['shape'], r)
"; done
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
SPMF_ROW_DYNAMIC=0 python bench.py --no-cpu-baseline --no-extras > $o/bench_static.json 2> $o/bench_static.err || { tail -5 $o/bench_static.err; exit 1; }
python3 -c "
import json
for f in ('bench','bench_static'):
    d=json.load(open('$o/'+f+'.json')); print(f, d['ms_per_step'], d['kernel_ms'])
"
