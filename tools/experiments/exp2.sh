set -x
mkdir -p gpurun_out/e2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/e2/parity_main.log 2>&1; tail -3 gpurun_out/e2/parity_main.log
SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_rowbal.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/e2/parity_rowbal.log 2>&1; tail -3 gpurun_out/e2/parity_rowbal.log
for v in main rowbal narrow wide4; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e2/$v.json 2> gpurun_out/e2/$v.err || tail -5 gpurun_out/e2/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e2/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
