set -x
mkdir -p gpurun_out/e19
python -m pytest tests/test_gpu_rule_and_surface.py -q -m gpu > gpurun_out/e19/tests.log 2>&1; tail -5 gpurun_out/e19/tests.log | cut -c1-400
