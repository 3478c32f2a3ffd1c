#!/bin/bash
# round 5, call H: the one-launch sampler / transform (second form) -- its tests and the A/B against the three launches
export TMPDIR=/tmp
o=gpurun_out/r5j
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_driver.py tests/test_gpu_p2p.py tests/test_gpu_deterministic.py -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
python tools/vi_fused_ab.py > $o/vi_fused_ab.jsonl 2> $o/vi_fused_ab.err || { tail -5 $o/vi_fused_ab.err; exit 1; }
cat $o/vi_fused_ab.jsonl
