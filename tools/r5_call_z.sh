#!/bin/bash
# round 5, second session: fit() on C4 and C5, the streamed-minibatch epoch (PCIe inclusive) and the resident minibatch
# epoch on the final library
export TMPDIR=/tmp
o=gpurun_out/r5e
mkdir -p $o
timeout -k 10 500 python tools/fit_c4.py 30 > $o/fit_c4.json 2> $o/fit_c4.err; echo "fit_c4 rc=$?"; cut -c1-700 $o/fit_c4.json
timeout -k 10 300 python tools/fit_c5.py 60 > $o/fit_c5.json 2> $o/fit_c5.err; echo "fit_c5 rc=$?"; cut -c1-700 $o/fit_c5.json
timeout -k 10 300 python tools/stream_epoch.py > $o/stream_epoch.json 2> $o/stream_epoch.err; echo "stream rc=$?"; cut -c1-900 $o/stream_epoch.json
timeout -k 10 300 python tools/minibatch_epoch.py > $o/minibatch_epoch.json 2> $o/minibatch_epoch.err; echo "minibatch rc=$?"; cut -c1-700 $o/minibatch_epoch.json
