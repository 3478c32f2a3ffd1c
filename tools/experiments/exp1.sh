set -x
mkdir -p gpurun_out/e1
for v in base rowpred colpred bothpred; do
  if [ $v = base ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e1/$v.json 2> gpurun_out/e1/$v.err || tail -5 gpurun_out/e1/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e1/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
tools/bin/gather_ceiling 100000000 0.016,0.064,0.25 > gpurun_out/e1/tiny_tables.txt 2>&1
cat gpurun_out/e1/tiny_tables.txt
