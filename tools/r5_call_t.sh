#!/bin/bash
# round 5, call T: the whole GPU suite on the current sources
export TMPDIR=/tmp
o=gpurun_out/r5u
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -5 $o/tests_gpu.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error" $o/tests_gpu.log | tail; exit 1; }
