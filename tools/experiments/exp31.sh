set -x
mkdir -p gpurun_out/e31
for pr in 4096 6144 8192 12288 16384; do
  python bench.py --panel-rows $pr --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/e31/p$pr.json 2> gpurun_out/e31/p$pr.err || tail -5 gpurun_out/e31/p$pr.err
  python -c "
import json;d=json.load(open('gpurun_out/e31/p$pr.json'));print('panel_rows $pr', round(d['ms_per_step'],4), d['kernel_ms'])"
done
