#!/bin/bash
# round 5, call A: GPU tests on the ABI-6 step, then the fixed-cost probe (new / legacy / FTD variants) and
# a kernel trace of the new step
export TMPDIR=/tmp
o=gpurun_out/r5b
mkdir -p $o
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $o/tests.log 2>&1; echo "tests rc=$?" > $o/tests.rc
tail -3 $o/tests.log
python tools/fixed_cost_probe.py 3 > $o/fixed_new.jsonl 2> $o/fixed_new.err || tail -5 $o/fixed_new.err
SPMF_LEGACY_STEP=1 python tools/fixed_cost_probe.py 3 > $o/fixed_legacy.jsonl 2> $o/fixed_legacy.err || tail -5 $o/fixed_legacy.err
for v in ftd16 ftd64; do
  SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so python tools/fixed_cost_probe.py 2 > $o/fixed_$v.jsonl 2> $o/fixed_$v.err || tail -5 $o/fixed_$v.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_new -- python3 tools/fixed_cost_probe.py 1 > $o/trace_new.jsonl 2> $o/trace_new.err || tail -5 $o/trace_new.err
for v in ftd16 ftd64; do
SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$v -- python3 tools/fixed_cost_probe.py 1 > $o/trace_$v.jsonl 2> $o/trace_$v.err || tail -5 $o/trace_$v.err
done
cat $o/fixed_new.jsonl $o/fixed_legacy.jsonl
