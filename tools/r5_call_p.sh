#!/bin/bash
# round 5, call P: a context keeps its p2p region (allocated once, re-used by later communicators): the p2p tests with
# re-created communicators at world 2 and 4, the sweep inside one set of processes, the counter
# traffic stamp of these sources, the default bench line
export TMPDIR=/tmp
o=gpurun_out/r5r
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py -q > $o/tests_p2p.log 2>&1
rc=$?; echo "p2p tests rc=$rc"; tail -3 $o/tests_p2p.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error|gave" $o/tests_p2p.log | tail; exit 1; }
timeout -k 10 700 python tools/p2p_sweep.py --same-processes > $o/p2p_sweep_same.jsonl 2> $o/p2p_sweep_same.err
grep -v "Gloo\|Feature\|amdgpu\|socket\|^$" $o/p2p_sweep_same.err | tail -5
cat $o/p2p_sweep_same.jsonl
bash tools/pmc.sh $o/pmc --workload c3 > $o/pmc_summary.txt 2> $o/pmc.err || { tail -5 $o/pmc.err; exit 1; }
python3 tools/pmc_traffic.py $o/pmc c3 $o/pmc_traffic.json && cp $o/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); print('C3', d['value'], d['ms_per_step'], d['roofline'])
"
