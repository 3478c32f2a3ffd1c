set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e4
python -m pytest tests -x -q -m gpu > gpurun_out/e4/gpu_tests.log 2>&1; tail -15 gpurun_out/e4/gpu_tests.log
python bench.py > gpurun_out/e4/bench_c3.json 2> gpurun_out/e4/bench_c3.err || tail -5 gpurun_out/e4/bench_c3.err
python -c "
import json;d=json.load(open('gpurun_out/e4/bench_c3.json'));print('c3', round(d['ms_per_step'],4), d['kernel_ms'], d['also'])"
python tools/notebook_pins.py noise 200 > gpurun_out/e4/pin_noise.json 2> gpurun_out/e4/pin_noise.err; cat gpurun_out/e4/pin_noise.json
python tools/notebook_pins.py linear 200 > gpurun_out/e4/pin_linear.json 2> gpurun_out/e4/pin_linear.err; cat gpurun_out/e4/pin_linear.json
