set -x
mkdir -p gpurun_out/e44
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "packed or canonical" > gpurun_out/e44/tests.log 2>&1 || { tail -30 gpurun_out/e44/tests.log; exit 1; }
tail -1 gpurun_out/e44/tests.log
python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e44/c3.json 2> gpurun_out/e44/c3.err || tail -5 gpurun_out/e44/c3.err
python -c "
import json;d=json.load(open('gpurun_out/e44/c3.json'));print(round(d['ms_per_step'],4), d['config'])"
