set -x
mkdir -p gpurun_out/e28
for v in main finabl1 finabl2 finabl3 finabl4; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --rows 125000 --no-cpu-baseline --no-extras --steps 100 --warmup 10 > gpurun_out/e28/$v.json 2> gpurun_out/e28/$v.err || tail -5 gpurun_out/e28/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e28/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
