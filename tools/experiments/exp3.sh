set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e3
python -m pytest tests -x -q -m gpu > gpurun_out/e3/gpu_tests.log 2>&1; tail -5 gpurun_out/e3/gpu_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e3/shard_prof -- python3 bench.py --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/e3/shard_bench.json 2> gpurun_out/e3/shard_prof.err || tail -5 gpurun_out/e3/shard_prof.err
f=$(find gpurun_out/e3/shard_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e3/shard_kernel_stats.csv; head -30 gpurun_out/e3/shard_kernel_stats.csv
rm -rf gpurun_out/e3/shard_prof
python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e3/c4.json 2> gpurun_out/e3/c4.err || tail -5 gpurun_out/e3/c4.err
python bench.py --workload c5 --no-cpu-baseline --no-extras --steps 10 --warmup 2 > gpurun_out/e3/c5.json 2> gpurun_out/e3/c5.err || tail -5 gpurun_out/e3/c5.err
for v in c4 c5; do python -c "
import json;d=json.load(open('gpurun_out/e3/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"; done
