set -x
mkdir -p gpurun_out/e47
python tools/fit_c3.py 200 1 > gpurun_out/e47/fit_c3.json 2> gpurun_out/e47/fit_c3.err || tail -20 gpurun_out/e47/fit_c3.err
cat gpurun_out/e47/fit_c3.json
