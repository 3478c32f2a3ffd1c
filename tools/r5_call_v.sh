#!/bin/bash
# round 5, call V: kernel trace of the K = 128 step (where do 2 ms per step go that the taps do not see?)
export TMPDIR=/tmp
o=$PWD/gpurun_out/r5v
mkdir -p $o
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/prof128 -- python3 $GRAFT_REPO_ROOT/tools/widek_probe.py only128 > $o/p128.out 2> $o/p128.err
echo "rc=$?"
cd $GRAFT_REPO_ROOT
cut -c1-250 $o/p128.out
f=$(find $o/prof128 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-220
