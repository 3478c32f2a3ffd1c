set -x
mkdir -p gpurun_out/e25
export SPMF_ROW_SPLIT=1
export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_m1lds.so
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e25/c3.json 2> gpurun_out/e25/c3.err || tail -5 gpurun_out/e25/c3.err
python -c "
import json;d=json.load(open('gpurun_out/e25/c3.json'));print('c3 split m1lds', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
unset SPMF_ROW_SPLIT
python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e25/c4.json 2> gpurun_out/e25/c4.err || tail -5 gpurun_out/e25/c4.err
python -c "
import json;d=json.load(open('gpurun_out/e25/c4.json'));print('c4 m1lds', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
