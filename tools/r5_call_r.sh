#!/bin/bash
# round 5, call R: latent dimensions above 64 (csrc/widek.hip; prep / finish / dense_ll at KP = 128, 256)
export TMPDIR=/tmp
o=gpurun_out/r5s
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_widek.py -x -q > $o/tests_widek.log 2>&1
rc=$?; echo "widek tests rc=$rc"; tail -30 $o/tests_widek.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_and_surface.py tests/test_gpu_driver.py -x -q > $o/tests_parity.log 2>&1
rc=$?; echo "parity tests rc=$rc"; tail -3 $o/tests_parity.log
