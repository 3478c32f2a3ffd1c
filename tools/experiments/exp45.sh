set -x
mkdir -p gpurun_out/e45
python -m pytest tests/test_gpu_bernoulli.py tests/test_gpu_mixed.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/e45/tests.log 2>&1 || { tail -30 gpurun_out/e45/tests.log; exit 1; }
tail -1 gpurun_out/e45/tests.log
for v in base main base main; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --workload c5 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > gpurun_out/e45/c5$v.json 2> gpurun_out/e45/c5$v.err || tail -5 gpurun_out/e45/c5$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e45/c5$v.json'));print('c5 $v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
