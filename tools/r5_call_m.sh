#!/bin/bash
# round 5, call M: the p2p tests with the column-split step over the peer kernel; the workgroup-count sweep of the collective on one GPU
export TMPDIR=/tmp
o=gpurun_out/r5o
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
timeout -k 10 500 python tools/p2p_sweep.py > $o/p2p_sweep.jsonl 2> $o/p2p_sweep.err || tail -5 $o/p2p_sweep.err
cat $o/p2p_sweep.jsonl
