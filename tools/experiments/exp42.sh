set -x
mkdir -p gpurun_out/e42
python -m pytest tests -q -m gpu -x > gpurun_out/e42/tests.log 2>&1 || { tail -30 gpurun_out/e42/tests.log; exit 1; }
tail -1 gpurun_out/e42/tests.log
for v in off on; do
  if [ $v = off ]; then export SPMF_PACKED_ENTRIES=0; else unset SPMF_PACKED_ENTRIES; fi
  for w in c3 c4 c5 c2; do
  python bench.py --workload $w --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e42/$w$v.json 2> gpurun_out/e42/$w$v.err || tail -5 gpurun_out/e42/$w$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e42/$w$v.json'));print('$w packed $v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
  done
done
