set -x
mkdir -p gpurun_out/e22
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bernoulli.py tests/test_gpu_mixed.py -m gpu -x -q > gpurun_out/e22/pytest.log 2>&1 || { tail -30 gpurun_out/e22/pytest.log; exit 1; }
tail -2 gpurun_out/e22/pytest.log
for v in main ftd16 ftd8; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e22/$v.json 2> gpurun_out/e22/$v.err || tail -5 gpurun_out/e22/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e22/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
