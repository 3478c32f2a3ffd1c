set -x
mkdir -p gpurun_out/e15
python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e15/c4.json 2> gpurun_out/e15/c4.err || tail -5 gpurun_out/e15/c4.err
python -c "
import json;d=json.load(open('gpurun_out/e15/c4.json'));print('c4', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
python tools/b3_err.py > gpurun_out/e15/b3_err.txt 2>gpurun_out/e15/err0.log; cat gpurun_out/e15/b3_err.txt
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_configs.py -q -m gpu -k "bf16x3 or c4_slice" > gpurun_out/e15/tests.log 2>&1; tail -4 gpurun_out/e15/tests.log | cut -c1-300
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e15/c4prof -- python3 bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e15/c4_under_rocprof.json 2> gpurun_out/e15/c4prof.err || tail -5 gpurun_out/e15/c4prof.err
f=$(find gpurun_out/e15/c4prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e15/kernel_stats_c4.csv; rm -rf gpurun_out/e15/c4prof
f=$(find gpurun_out/e15 -name '*kernel_trace.csv' | head -1)
grep "expdot3" gpurun_out/e15/kernel_stats_c4.csv | cut -c1-40,200-330
