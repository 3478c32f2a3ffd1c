set -x
mkdir -p gpurun_out/e30
for v in serial overlap; do
  if [ $v = serial ]; then unset SPMF_DENSE_OVERLAP; else export SPMF_DENSE_OVERLAP=1; fi
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e30/$v.json 2> gpurun_out/e30/$v.err || tail -5 gpurun_out/e30/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e30/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
export SPMF_DENSE_OVERLAP=1
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/e30/tests.log 2>&1 || tail -30 gpurun_out/e30/tests.log
tail -2 gpurun_out/e30/tests.log
