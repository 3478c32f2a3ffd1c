set -x
mkdir -p gpurun_out/e51
python tools/fit_c3.py 200 1 > gpurun_out/e51/fit_c3.json 2> gpurun_out/e51/fit_c3.err || tail -20 gpurun_out/e51/fit_c3.err
cat gpurun_out/e51/fit_c3.json
python -m pytest tests/test_gpu_driver.py -q -m gpu -x > gpurun_out/e51/tests.log 2>&1 || tail -20 gpurun_out/e51/tests.log
tail -1 gpurun_out/e51/tests.log
