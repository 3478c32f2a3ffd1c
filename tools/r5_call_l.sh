#!/bin/bash
# round 5, call L: randomised parity against the fp64 oracle on the round's final library (new seeds), the layout builder
# against its torch statement, fit() on the whole C3 matrix, eight C3s of rows as one shard
export TMPDIR=/tmp
o=gpurun_out/r5n
mkdir -p $o
( while true; do sleep 60; echo tick >> $o/alive.txt; done ) &
tick=$!
python tools/stress_parity.py 300 9101 > $o/stress_linear.txt 2>&1; echo "rc $?" >> $o/stress_linear.txt; tail -3 $o/stress_linear.txt
python tools/stress_parity_modes.py 100 9102 > $o/stress_modes.txt 2>&1; echo "rc $?" >> $o/stress_modes.txt; tail -3 $o/stress_modes.txt
python tools/stress_layout.py 400 9103 > $o/stress_layout.txt 2>&1; echo "rc $?" >> $o/stress_layout.txt; tail -2 $o/stress_layout.txt
python tools/fit_c3.py 200 1 > $o/fit_c3.json 2> $o/fit_c3.err; tail -1 $o/fit_c3.json | cut -c1-400
python bench.py --rows 8000000 --no-cpu-baseline --steps 5 --warmup 2 > $o/bench_c3x8.json 2> $o/bench_c3x8.err; python3 -c "
import json
d=json.load(open('$o/bench_c3x8.json')); print('8M rows:', d['ms_per_step'], d['kernel_ms'], d['also'].get('layout_build_ms'))
"
kill $tick
