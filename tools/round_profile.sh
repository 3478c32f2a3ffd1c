#!/bin/bash
# One profiling pass of the headline workload for profiles/: rocprofv3 kernel stats of
# the bench command, the separate PMC passes, and the stamped traffic file.
# usage: tools/round_profile.sh <tag>        (e.g. r02)   -- run on the GPU box
set -e
export TMPDIR=/tmp
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
bash tools/pmc.sh $out/pmc --workload c3 > $out/${tag}_pmc_summary.txt
python3 tools/pmc_traffic.py $out/pmc c3 $out/pmc_traffic.json > /dev/null
head -12 $out/${tag}_kernel_stats.csv
