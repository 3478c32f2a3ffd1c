set -x
mkdir -p gpurun_out/e37
export SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --rows 250000 --steps 3 --warmup 1 > gpurun_out/e37/c3_2rank_gloo.json 2> gpurun_out/e37/c3_2rank.err || tail -20 gpurun_out/e37/c3_2rank.err
unset SPMF_BENCH_BACKEND SPMF_BENCH_ONE_GPU
python bench.py --rows 250000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/e37/c3_1rank.json 2> gpurun_out/e37/c3_1rank.err
python -c "
import json
a=json.load(open('gpurun_out/e37/c3_2rank_gloo.json')); b=json.load(open('gpurun_out/e37/c3_1rank.json'))
print('2rank', a['n_gpus'], a['ms_per_step'], a['elbo_x'], a['config']['rows'], a['config']['nnz']); print('1rank', b['ms_per_step'], b['elbo_x'], b['config']['nnz'])
print('rel diff', abs(a['elbo_x']-b['elbo_x'])/abs(b['elbo_x']))"
