#!/bin/bash
# round 5, second session: kernel statistics of the C4 and C5 lines on the final library
export TMPDIR=/tmp
o=gpurun_out/r5c
mkdir -p $o
for w in c4 c5 c2; do
  st=20; [ $w = c4 ] && st=5
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_$w -- python3 bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline --no-extras > $o/bench_${w}_under_rocprof.json 2> $o/stats_$w.err || { tail -5 $o/stats_$w.err; exit 1; }
  f=$(find $o/stats_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { cp "$f" $o/kernel_stats_$w.csv; head -7 "$f" | cut -c1-160; }
  rm -rf $o/stats_$w
  timeout -k 10 300 python bench.py --workload $w --steps $st --no-cpu-baseline --no-extras > $o/bench_$w.json 2> $o/bench_$w.err || { tail -5 $o/bench_$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$o/bench_$w.json')); print('$w', round(d['ms_per_step'],4), d['kernel_ms'], d.get('roofline'))
"
done
