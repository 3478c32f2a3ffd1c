#!/bin/bash
# round 5, call Q: padded gather slots dropped by the buffer range check (common.h GTable, SPMF_OOB_PAD) --
# parity subset on the new library, then A/B of the C3 / C2 / C4 / C5 lines against the variant built with
# -DSPMF_OOB_PAD=0 (padded slots read row 0 as before)
export TMPDIR=/tmp
o=gpurun_out/r5q
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_deterministic.py tests/test_gpu_mixed.py tests/test_gpu_logtransform.py -x -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $o/tests.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error" $o/tests.log | tail; exit 1; }
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_oobpad0.so; else unset SPMF_LIB_PATH; fi
  for w in c3 c2 c5 c4; do
    st=20; [ $w = c4 ] && st=4
    [ $rep = 2 ] && [ $w != c3 ] && continue
    timeout -k 10 300 python bench.py --workload $w --steps $st --no-extras --no-cpu-baseline > $o/bench_${w}_${v}_$rep.json 2> $o/bench_${w}_${v}_$rep.err || { tail -5 $o/bench_${w}_${v}_$rep.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$o/bench_${w}_${v}_$rep.json')); print('$w $v $rep', round(d['ms_per_step'],4), d['kernel_ms'], d.get('elbo_x'))
"
  done
done
done
