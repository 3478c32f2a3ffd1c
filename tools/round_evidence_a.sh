# GPU box: full parity suite, the default bench line, kernel stats + PMC passes of the same command
set -x
export TMPDIR=/tmp
tag=${1:-rXX}
out=gpurun_out/ev_$tag
mkdir -p $out
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -3 $out/gpu_tests.log
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
python -c "
import json;d=json.load(open('$out/bench_c3.json'));print('c3', round(d['ms_per_step'],4), d['kernel_ms'], d['roofline'], d['cpu_baseline'], d['also'])"
bash tools/round_profile.sh $tag
