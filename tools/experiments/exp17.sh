set -x
mkdir -p gpurun_out/e17
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_launch.py tests/test_gpu_driver.py -q -m gpu > gpurun_out/e17/tests.log 2>&1; tail -5 gpurun_out/e17/tests.log | cut -c1-300
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/e17/r03_bench_c4.json 2> gpurun_out/e17/c4.err || tail -5 gpurun_out/e17/c4.err
python -c "
import json;d=json.load(open('gpurun_out/e17/r03_bench_c4.json'));print('c4', round(d['ms_per_step'],4), d['kernel_ms'], d['roofline'], d['also'])"
timeout 300 python tools/stress_parity.py 60 4242 > gpurun_out/e17/stress.txt 2>&1; tail -6 gpurun_out/e17/stress.txt | cut -c1-300
