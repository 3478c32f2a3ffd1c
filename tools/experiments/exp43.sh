set -x
mkdir -p gpurun_out/e43
for v in main rowblk main rowblk; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > gpurun_out/e43/$v.json 2> gpurun_out/e43/$v.err || tail -5 gpurun_out/e43/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e43/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
