set -x
mkdir -p gpurun_out/e10
python tools/b3_err.py > gpurun_out/e10/b3_err.txt 2>gpurun_out/e10/err0.log; cat gpurun_out/e10/b3_err.txt
for v in main nw8pb1; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e10/c4_$v.json 2> gpurun_out/e10/c4_$v.err || tail -5 gpurun_out/e10/c4_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e10/c4_$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
unset SPMF_LIB_PATH
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_configs.py -q -m gpu -k "bf16x3 or c4_slice" > gpurun_out/e10/tests.log 2>&1; tail -4 gpurun_out/e10/tests.log | cut -c1-300
