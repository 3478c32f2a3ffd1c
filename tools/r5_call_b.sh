#!/bin/bash
# round 5, call B: the peer-pointer collective at world 2 / 4, parity subset after the finish-body changes,
# and where the begin kernel's time goes (ablations of the prior half: timing only)
export TMPDIR=/tmp
o=gpurun_out/r5c
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py -x -q > $o/p2p.log 2>&1; echo "p2p rc=$?" > $o/p2p.rc
tail -15 $o/p2p.log
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_and_surface.py tests/test_gpu_bernoulli.py -x -q -m gpu > $o/subset.log 2>&1; echo "subset rc=$?" > $o/subset.rc
tail -3 $o/subset.log
python tools/fixed_cost_probe.py 2 > $o/fixed_new.jsonl 2> $o/fixed_new.err || tail -5 $o/fixed_new.err
for v in abl1 abl2 abl4 abl8 abl15; do
  SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so python tools/fixed_cost_probe.py 1 > $o/fixed_$v.jsonl 2> $o/fixed_$v.err || tail -5 $o/fixed_$v.err
done
for f in $o/fixed_*.jsonl; do echo $f; python3 -c "
import json,sys
for l in open('$f'):
    d=json.loads(l); r=d['runs'][-1]; print('  ', d['shape'], 'step', r['ms_no_taps'], 'begin', r['prep'], 'end', r['finish'])
"; done
