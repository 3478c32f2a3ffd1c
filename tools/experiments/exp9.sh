set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e9
python -m pytest tests -q -m gpu > gpurun_out/e9/gpu_tests.log 2>&1; tail -6 gpurun_out/e9/gpu_tests.log | cut -c1-300
python bench.py > gpurun_out/e9/r03_bench_c3.json 2> gpurun_out/e9/bench_c3.err || tail -5 gpurun_out/e9/bench_c3.err
bash tools/round_profile.sh r03 > gpurun_out/e9/round_profile.log 2>&1; tail -14 gpurun_out/e9/round_profile.log | cut -c1-200
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/e9/r03_bench_c4.json 2> gpurun_out/e9/c4.err || tail -5 gpurun_out/e9/c4.err
SPMF_DENSE_BF16X3=0 python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 > gpurun_out/e9/r03_bench_c4_f32mfma.json 2> gpurun_out/e9/c4f.err || tail -5 gpurun_out/e9/c4f.err
python bench.py --workload c5 --no-cpu-baseline > gpurun_out/e9/r03_bench_c5.json 2> gpurun_out/e9/c5.err || tail -5 gpurun_out/e9/c5.err
python bench.py --workload c2 --no-cpu-baseline --no-extras > gpurun_out/e9/r03_bench_c2.json 2> gpurun_out/e9/c2.err || tail -5 gpurun_out/e9/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e9/c4prof -- python3 bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e9/r03_bench_c4_under_rocprof.json 2> gpurun_out/e9/c4prof.err || tail -5 gpurun_out/e9/c4prof.err
f=$(find gpurun_out/e9/c4prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e9/r03_kernel_stats_c4.csv; rm -rf gpurun_out/e9/c4prof
for v in r03_bench_c3 r03_bench_c4 r03_bench_c4_f32mfma r03_bench_c5 r03_bench_c2; do python -c "
import json;d=json.load(open('gpurun_out/e9/$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'], d['roofline']['frac'], d.get('saturated'), {k:v for k,v in d['also'].items() if 'after50' in k})"; done
