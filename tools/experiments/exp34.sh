set -x
mkdir -p gpurun_out/e34
python -m pytest tests -q -m gpu > gpurun_out/e34/gpu_tests.log 2>&1 || { tail -40 gpurun_out/e34/gpu_tests.log; exit 1; }
tail -2 gpurun_out/e34/gpu_tests.log
python bench.py > gpurun_out/e34/c3.json 2> gpurun_out/e34/c3.err || { tail -5 gpurun_out/e34/c3.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/e34/c3.json'));print('c3', round(d['ms_per_step'],4), d['kernel_ms'], d['config'], d['also'])"
for w in c4 c5 c2; do
  python bench.py --workload $w > gpurun_out/e34/$w.json 2> gpurun_out/e34/$w.err || { tail -5 gpurun_out/e34/$w.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/e34/$w.json'));print('$w', round(d['ms_per_step'],4), d['kernel_ms'], d['config'].get('panel_rows'))"
done
