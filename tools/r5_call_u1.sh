#!/bin/bash
# round 5, final evidence 1: smoke, the whole GPU suite, the default bench line, the kernel statistics of the same
# command under rocprofv3, the wide-K probe -- all on the final sources
export TMPDIR=/tmp
o=$PWD/gpurun_out/r5x
mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || { tail -20 $o/smoke.log; exit 1; }
tail -1 $o/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > $o/tests_gpu.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -3 $o/tests_gpu.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error" $o/tests_gpu.log | tail; exit 1; }
( time python bench.py ) > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
tail -4 $o/bench.err
python3 -c "
import json
d=json.load(open('$o/bench.json')); print('C3', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']); a=d['also']
print({k:a[k] for k in a if k.startswith('k128') or k in ('c2_ms_per_step','c4_ms_per_step','c5_ms_per_step','shard125k_ms_per_step_no_taps','minibatch_ms_per_step','c1_gpu_ms_per_step','after50_ms_per_step','vi_step_ms')})
print('vi', d.get('vi_step_ms'))
"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $o/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 25 > $o/bench_under_rocprof.json 2> $o/prof.err
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp "$f" $o/kernel_stats_c3.csv; head -8 "$f" | cut -c1-200; }
timeout -k 10 300 python tools/widek_probe.py > $o/widek_probe.jsonl 2> $o/widek_probe.err
cut -c1-240 $o/widek_probe.jsonl
rm -rf $o/prof
