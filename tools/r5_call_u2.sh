#!/bin/bash
# round 5, final evidence 2: kernel statistics (csv) of the bench command, the traffic counters (stamped with the
# kernel sources' hash: bench.py's roofline.traffic), the request-path counters of the two sparse passes
export TMPDIR=/tmp
o=gpurun_out/r5y
mkdir -p $o
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-extras > $o/bench_under_rocprof.json 2> $o/stats.err || { tail -5 $o/stats.err; exit 1; }
f=$(find $o/stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { cp "$f" $o/kernel_stats_c3.csv; head -8 "$f" | cut -c1-220; }
echo "--- pmc.sh"
bash tools/pmc.sh $o/pmc --workload c3 > $o/pmc_summary.txt 2> $o/pmc.err || { tail -5 $o/pmc.err; exit 1; }
python3 tools/pmc_traffic.py $o/pmc c3 $o/pmc_traffic.json && cat $o/pmc_traffic.json | head -30
echo "--- pmc_r05.sh"
bash tools/pmc_r05.sh $o/pmc5 > $o/pmc5.log 2>&1 || { tail -5 $o/pmc5.log; exit 1; }
python3 tools/pmc_r05_report.py $o/pmc5 > $o/pmc_sparse_passes.txt 2> $o/report.err || tail -3 $o/report.err
grep -A8 "derived, vector" $o/pmc_sparse_passes.txt | head -40
# keep what is merged back small: the raw rocprofv3 output directories stay on the box
rm -rf $o/stats $o/pmc/p*/ $o/pmc5/p*/
du -sh $o | tail -1
