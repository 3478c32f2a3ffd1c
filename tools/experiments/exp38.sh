set -x
mkdir -p gpurun_out/e38
export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_rowblk.so
python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/e38/tests.log 2>&1 || { tail -20 gpurun_out/e38/tests.log; exit 1; }
tail -1 gpurun_out/e38/tests.log
for v in main rowblk; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e38/$v.json 2> gpurun_out/e38/$v.err || tail -5 gpurun_out/e38/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e38/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e38/c4$v.json 2> gpurun_out/e38/c4$v.err || tail -5 gpurun_out/e38/c4$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e38/c4$v.json'));print('c4 $v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
