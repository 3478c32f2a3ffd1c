set -x
mkdir -p gpurun_out/e14
for v in abl1 abl2; do
  export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so
  python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/e14/c4_$v.json 2> gpurun_out/e14/c4_$v.err || tail -5 gpurun_out/e14/c4_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e14/c4_$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
