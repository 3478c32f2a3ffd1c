#!/bin/bash
# round 5, call E: how a 128-byte row is asked for (halves probe, plain and under counters), the dynamic share of the
# row pass (1/8 default against 1/4, 1/2, 7/8), the encode-only sweep's L2 occupancy
export TMPDIR=/tmp
o=gpurun_out/r5g
mkdir -p $o
tools/bin/gather_halves_probe > $o/halves.txt 2> $o/halves.err || { tail -3 $o/halves.err; exit 1; }
cat $o/halves.txt
i=0
for ctrs in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_HIT_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $o/halves_pmc/p$i -- tools/bin/gather_halves_probe 40000000 > $o/halves_pmc_p$i.out 2> $o/halves_pmc_p$i.err
  rc=$?; [ $rc -lt 124 ] || { echo "halves pmc pass $i killed"; exit 1; }
done
for v in main dyn1_4 dyn1_2 dyn7_8; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras > $o/bench_$v.json 2> $o/bench_$v.err || { tail -3 $o/bench_$v.err; exit 1; }
  python tools/fixed_cost_probe.py 2 > $o/fixed_$v.jsonl 2> $o/fixed_$v.err || { tail -3 $o/fixed_$v.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$o/bench_$v.json')); print('$v', 'C3', round(d['ms_per_step'],4), d['kernel_ms']['row_pass'], d['kernel_ms']['col_pass'])
for l in open('$o/fixed_$v.jsonl'):
    e=json.loads(l); r=e['runs'][-1]; print('   ', e['shape'], r['ms_no_taps'], 'row', r['row'])
"
done
unset SPMF_LIB_PATH
timeout -k 5 240 rocprofv3 --pmc TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_HIT_sum --kernel-trace --output-format csv -d $o/enc/p1 -- python3 tools/encode_only_loop.py > $o/enc_p1.out 2> $o/enc_p1.err
python3 tools/pmc_r05_summary.py $o/enc > $o/enc_summary.txt 2>&1; grep -A6 "row_pass" $o/enc_summary.txt | head -10
