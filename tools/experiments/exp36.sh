set -x
mkdir -p gpurun_out/e36
for v in main colw5 colw3; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e36/$v.json 2> gpurun_out/e36/$v.err || tail -5 gpurun_out/e36/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e36/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
unset SPMF_LIB_PATH
for pr in 9472 10432 13920 15648; do
  python bench.py --panel-rows $pr --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e36/p$pr.json 2> gpurun_out/e36/p$pr.err || tail -5 gpurun_out/e36/p$pr.err
  python -c "
import json;d=json.load(open('gpurun_out/e36/p$pr.json'));print('panel_rows $pr', round(d['ms_per_step'],4), d['kernel_ms'])"
done
