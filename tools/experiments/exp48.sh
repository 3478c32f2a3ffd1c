set -x
mkdir -p gpurun_out/e48
python -m pytest tests/test_gpu_bernoulli.py tests/test_gpu_mixed.py tests/test_gpu_configs.py tests/test_gpu_two_rank.py -q -m gpu -x > gpurun_out/e48/tests.log 2>&1 || { tail -30 gpurun_out/e48/tests.log; exit 1; }
tail -1 gpurun_out/e48/tests.log
for i in 1 2; do
python bench.py --workload c5 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > gpurun_out/e48/c5_$i.json 2> gpurun_out/e48/c5_$i.err || tail -5 gpurun_out/e48/c5_$i.err
python -c "
import json;d=json.load(open('gpurun_out/e48/c5_$i.json'));print('c5', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
