#!/bin/bash
# round 5, call S: row pass with the first chunk's V' gathers issued before sweep 1 (ROW_V_AHEAD) --
# parity subset, then A/B against variants named on the command line (default: vahead0)
export TMPDIR=/tmp
o=gpurun_out/r5t
mkdir -p $o
timeout -k 10 900 python -m pytest ${SPMF_CALL_TESTS:-tests/test_gpu_parity.py} -x -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $o/tests.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error" $o/tests.log | tail; exit 1; }
variants=${@:-vahead0}
for rep in 1 2; do
for v in new $variants; do
  if [ $v = new ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  for w in c3 c2 c5; do
    [ $rep = 2 ] && [ $w != c3 ] && continue
    timeout -k 10 300 python bench.py --workload $w --steps 20 --no-extras --no-cpu-baseline > $o/bench_${w}_${v}_$rep.json 2> $o/bench_${w}_${v}_$rep.err || { tail -5 $o/bench_${w}_${v}_$rep.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$o/bench_${w}_${v}_$rep.json')); print('$w $v $rep', round(d['ms_per_step'],4), d['kernel_ms'], d.get('elbo_x'))
"
  done
done
done
