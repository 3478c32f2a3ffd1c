# GPU box, round 4 (a): full parity suite, the default bench line, kernel stats + PMC passes of the same command
set -x
export TMPDIR=/tmp
tag=r04
out=gpurun_out/ev_$tag
mkdir -p $out
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -3 $out/gpu_tests.log
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
bash tools/round_profile.sh $tag
