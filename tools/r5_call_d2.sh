#!/bin/bash
# round 5, call D2: counters -- the L2 / SQ / GRBM groups of the sparse passes (TCP groups: call C), the encode-only
# sweep under the TCP groups; a kernel trace of the VI step with the hierarchy on the side stream; the K-halving proxy on C4
export TMPDIR=/tmp
o=gpurun_out/r5f
mkdir -p $o/enc
SKIP_PASSES="1 2 3" tools/pmc_r05.sh $o/pmc > $o/pmc.log 2>&1 || { tail -3 $o/pmc.log; exit 1; }
tail -3 $o/pmc.log
i=0
for ctrs in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $o/enc/p$i -- python3 tools/encode_only_loop.py > $o/enc/p$i.out 2> $o/enc/p$i.err
  rc=$?; [ $rc -lt 124 ] || { echo "encode pass $i killed"; exit 1; }
done
python3 tools/pmc_r05_summary.py $o/enc > $o/enc_summary.txt 2>&1; grep -A8 "row_pass" $o/enc_summary.txt | head -24
timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $o/vitrace -- python3 tools/vi_overlap_trace.py > $o/vitrace.out 2> $o/vitrace.err
rc=$?; [ $rc -lt 124 ] || { echo "vi trace killed"; exit 1; }
python3 tools/vi_overlap_trace.py --analyze $o/vitrace > $o/vitrace_step.txt 2>&1; cat $o/vitrace_step.txt
python tools/c4_khalf_probe.py > $o/c4_khalf.json 2> $o/c4_khalf.err || tail -5 $o/c4_khalf.err
cat $o/c4_khalf.json
