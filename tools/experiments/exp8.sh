set -x
mkdir -p gpurun_out/e8
echo "== bf16x3 pipelined (default)" > gpurun_out/e8/b3_err.txt; python tools/b3_err.py >> gpurun_out/e8/b3_err.txt 2>gpurun_out/e8/err0.log
cat gpurun_out/e8/b3_err.txt
python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/e8/c4.json 2> gpurun_out/e8/c4.err || tail -5 gpurun_out/e8/c4.err
python -c "
import json;d=json.load(open('gpurun_out/e8/c4.json'));print('c4', round(d['ms_per_step'],4), d['kernel_ms'], d['roofline'])"
python -m pytest tests/test_gpu_logtransform.py tests/test_gpu_configs.py -q -m gpu -k "bf16x3 or c4_slice or log_transform" > gpurun_out/e8/tests.log 2>&1; tail -5 gpurun_out/e8/tests.log | cut -c1-400
python tools/col_order_probe_c4.py > gpurun_out/e8/col_order_c4.txt 2> gpurun_out/e8/col_order_c4.err; cat gpurun_out/e8/col_order_c4.txt
