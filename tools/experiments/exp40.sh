set -x
mkdir -p gpurun_out/e40
python -m pytest tests/test_gpu_parity.py tests/test_gpu_logtransform.py -q -m gpu -x > gpurun_out/e40/tests.log 2>&1 || { tail -30 gpurun_out/e40/tests.log; exit 1; }
tail -1 gpurun_out/e40/tests.log
for v in off on; do
  if [ $v = off ]; then export SPMF_PACKED_ENTRIES=0; else unset SPMF_PACKED_ENTRIES; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e40/c3$v.json 2> gpurun_out/e40/c3$v.err || tail -5 gpurun_out/e40/c3$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e40/c3$v.json'));print('c3 packed $v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
  python bench.py --workload c5 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e40/c5$v.json 2> gpurun_out/e40/c5$v.err || tail -5 gpurun_out/e40/c5$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e40/c5$v.json'));print('c5 packed $v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
