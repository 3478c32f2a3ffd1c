# GPU box: the round-end checks the driver runs -- smoke(), the GPU suite, the default bench line
set -x
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 || { tail -20 gpurun_out/final/smoke.log; exit 1; }
tail -1 gpurun_out/final/smoke.log
python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1 || { tail -40 gpurun_out/final/gpu_tests.log; exit 1; }
tail -2 gpurun_out/final/gpu_tests.log
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail -5 gpurun_out/final/bench.err; exit 1; }
tail -4 gpurun_out/final/bench.err
python -c "
import json;d=json.load(open('gpurun_out/final/bench.json'));print(round(d['ms_per_step'],4), d['value'], d['roofline'])"
