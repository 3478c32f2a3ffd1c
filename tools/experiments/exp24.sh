set -x
mkdir -p gpurun_out/e24
for v in main split; do
  if [ $v = main ]; then unset SPMF_ROW_SPLIT; else export SPMF_ROW_SPLIT=1; fi
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/e24/$v.json 2> gpurun_out/e24/$v.err || tail -5 gpurun_out/e24/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e24/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'], {k:v for k,v in d.get('also',{}).items() if 'shard125k_ms' in k or 'shard125k_kernel' in k})"
done
