set -x
mkdir -p gpurun_out/e21
for v in main rowgrp8; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e21/$v.json 2> gpurun_out/e21/$v.err || tail -5 gpurun_out/e21/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e21/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
