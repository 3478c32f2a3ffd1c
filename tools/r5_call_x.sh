#!/bin/bash
# round 5, second session: randomised parity against the fp64 oracle on the final library, new seeds
export TMPDIR=/tmp
o=gpurun_out/r5s2
mkdir -p $o
timeout -k 10 500 python tools/stress_parity.py 300 9201 > $o/stress_linear.txt 2> $o/stress_linear.err; echo "linear rc=$?"; tail -3 $o/stress_linear.txt
timeout -k 10 500 python tools/stress_parity.py 120 9202 widek > $o/stress_widek.txt 2> $o/stress_widek.err; echo "widek rc=$?"; tail -3 $o/stress_widek.txt
timeout -k 10 500 python tools/stress_parity_modes.py 100 9203 > $o/stress_modes.txt 2> $o/stress_modes.err; echo "modes rc=$?"; tail -3 $o/stress_modes.txt
timeout -k 10 300 python tools/stress_layout.py 300 9204 > $o/stress_layout.txt 2> $o/stress_layout.err; echo "layout rc=$?"; tail -2 $o/stress_layout.txt
