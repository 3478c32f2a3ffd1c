set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e6
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_configs.py tests/test_gpu_bernoulli.py tests/test_gpu_dense.py -q -m gpu > gpurun_out/e6/gpu_tests.log 2>&1; tail -30 gpurun_out/e6/gpu_tests.log | cut -c1-600
python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e6/c4_f32.json 2> gpurun_out/e6/c4_f32.err || tail -5 gpurun_out/e6/c4_f32.err
SPMF_DENSE_BF16X3=1 python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e6/c4_b3.json 2> gpurun_out/e6/c4_b3.err || tail -5 gpurun_out/e6/c4_b3.err
for v in c4_f32 c4_b3; do python -c "
import json;d=json.load(open('gpurun_out/e6/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'], d['roofline'])"; done
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 SPMF_BENCH_COMM=lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e6/rccl_prof -- python3 bench.py --gpus 1 --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/e6/rccl_bench.json 2> gpurun_out/e6/rccl_prof.err || tail -5 gpurun_out/e6/rccl_prof.err
f=$(find gpurun_out/e6/rccl_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e6/rccl_kernel_stats.csv; grep "spmf::\|ccl" gpurun_out/e6/rccl_kernel_stats.csv | awk -F'","|",' '{print substr($1,1,60), $2, $4}' | head -14
rm -rf gpurun_out/e6/rccl_prof
