set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e5
python -m pytest tests -q -m gpu > gpurun_out/e5/gpu_tests.log 2>&1; tail -40 gpurun_out/e5/gpu_tests.log
python bench.py > gpurun_out/e5/bench_c3.json 2> gpurun_out/e5/bench_c3.err || tail -5 gpurun_out/e5/bench_c3.err
python -c "
import json;d=json.load(open('gpurun_out/e5/bench_c3.json'));print('c3', round(d['ms_per_step'],4), d['kernel_ms'], d['also'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e5/shard_prof -- python3 bench.py --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/e5/shard_bench.json 2> gpurun_out/e5/shard_prof.err || tail -5 gpurun_out/e5/shard_prof.err
f=$(find gpurun_out/e5/shard_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e5/shard_kernel_stats.csv; grep "spmf::" gpurun_out/e5/shard_kernel_stats.csv | cut -c1-60,200-400 | head -12
rm -rf gpurun_out/e5/shard_prof
