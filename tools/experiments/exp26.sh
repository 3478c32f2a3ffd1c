set -x
mkdir -p gpurun_out/e26
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_bernoulli.py tests/test_gpu_mixed.py tests/test_gpu_dense.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/e26/pytest.log 2>&1 || { tail -30 gpurun_out/e26/pytest.log; exit 1; }
tail -2 gpurun_out/e26/pytest.log
for w in c4 c5; do
python bench.py --workload $w --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e26/$w.json 2> gpurun_out/e26/$w.err || tail -5 gpurun_out/e26/$w.err
python -c "
import json;d=json.load(open('gpurun_out/e26/$w.json'));print('$w', round(d['ms_per_step'],4), d['kernel_ms'], d['elbo_x'])"
done
