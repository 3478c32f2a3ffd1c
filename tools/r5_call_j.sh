#!/bin/bash
# round 5, call J: FETCH_SIZE / WRITE_SIZE / hit-rate passes on the final kernel sources -> profiles/pmc_traffic.json (roofline.traffic)
export TMPDIR=/tmp
o=gpurun_out/r5l
mkdir -p $o
bash tools/pmc.sh $o/pmc --workload c3 > $o/pmc_summary.txt 2> $o/pmc.err || { tail -5 $o/pmc.err; exit 1; }
python3 tools/pmc_traffic.py $o/pmc c3 $o/pmc_traffic.json
cat $o/pmc_traffic.json | head -60
