set -x
mkdir -p gpurun_out/e29
for v in main rowgrp2 rowgrp8 colgrp2 colgrp8; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e29/$v.json 2> gpurun_out/e29/$v.err || tail -5 gpurun_out/e29/$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e29/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
