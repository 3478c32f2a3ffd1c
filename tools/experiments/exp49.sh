set -x
bash tools/round_profile.sh r03 > gpurun_out/prof_again.txt 2>&1 || { tail -20 gpurun_out/prof_again.txt; exit 1; }
mkdir -p gpurun_out/e49
python bench.py --workload c5 > gpurun_out/e49/bench_c5.json 2> gpurun_out/e49/bench_c5.err || tail -5 gpurun_out/e49/bench_c5.err
python -c "
import json;d=json.load(open('gpurun_out/e49/bench_c5.json'));print('c5', round(d['ms_per_step'],4), d['kernel_ms'], d.get('also'))"
bash tools/final_check.sh
