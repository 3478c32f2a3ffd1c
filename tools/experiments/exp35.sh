set -x
mkdir -p gpurun_out/e35
for v in main rownt colnt; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e35/$v.json 2> gpurun_out/e35/$v.err || tail -5 gpurun_out/e35/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e35/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
for v in main rownt colnt; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e35/c4$v.json 2> gpurun_out/e35/c4$v.err || tail -5 gpurun_out/e35/c4$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e35/c4$v.json'));print('c4 $v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
