#!/bin/bash
# round 5, call C: the whole GPU suite on the rebuilt finish body, the fixed-cost probe, the bench line, then
# the request-path counters of the sparse passes.  Nothing runs after a failed or aborted step.
export TMPDIR=/tmp
o=gpurun_out/r5d
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1          # (failed assertions: go on; an abort, a timeout or a collection error: stop)
grep -E "^(FAILED|ERROR)" $o/tests.log
python tools/fixed_cost_probe.py 3 > $o/fixed_new.jsonl 2> $o/fixed_new.err || { tail -5 $o/fixed_new.err; exit 1; }
python3 -c "
import json
for l in open('$o/fixed_new.jsonl'):
    d=json.loads(l); r=d['runs'][-1]; print('  ', d['shape'], r)
"
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python tools/row_order_probe.py 1000000 > $o/row_order.json 2> $o/row_order.err || tail -5 $o/row_order.err
tools/pmc_r05.sh $o/pmc > $o/pmc.log 2>&1
python3 tools/pmc_r05_summary.py $o/pmc $o/pmc_summary.json > $o/pmc_summary.txt
tail -5 $o/pmc.log
