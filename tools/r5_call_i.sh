#!/bin/bash
# round 5, call I: the record -- whole GPU suite, smoke(), the default bench line, its kernel stats under rocprofv3
export TMPDIR=/tmp
o=gpurun_out/r5k
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || { tail -5 $o/smoke.log; exit 1; }
tail -1 $o/smoke.log
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); a=d['also']
print('C3', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], 'vi', d['vi_step_ms'])
for k in sorted(a):
    if any(s in k for s in ('shard125k_vi', 'shard125k_ms', 'shard125k_fixed', 'c1_gpu', 'ref_harness', 'c2_ms', 'c4_ms', 'c5_ms', 'minibatch_ms', 'S20', 'det_ms')): print('  ', k, a[k])
"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $o/bench_rocprof.json 2> $o/bench_rocprof.err
rc=$?; [ $rc -lt 124 ] || { echo "rocprof run killed"; exit 1; }
python3 - "$o" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'spmf' in r['Name']: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
