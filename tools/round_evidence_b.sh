# GPU box: shard profiles (plain and through the one-rank RCCL communicator), C4 / C5 / C2 bench lines, C4 kernel stats
set -x
export TMPDIR=/tmp
tag=${1:-rXX}
out=gpurun_out/ev_$tag
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "chunk_boundaries" > $out/new_tests.log 2>&1 || { tail -30 $out/new_tests.log; exit 1; }
tail -2 $out/new_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_prof -- python3 bench.py --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/shard_bench.json 2> $out/shard_prof.err || { tail -5 $out/shard_prof.err; exit 1; }
cp "$(find $out/shard_prof -name '*kernel_stats.csv' | head -1)" $out/${tag}_shard_kernel_stats.csv && rm -rf $out/shard_prof
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 SPMF_BENCH_COMM=lib rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_prof1 -- python3 bench.py --gpus 1 --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/shard_1rank_bench.json 2> $out/shard_prof1.err || { tail -5 $out/shard_prof1.err; exit 1; }
cp "$(find $out/shard_prof1 -name '*kernel_stats.csv' | head -1)" $out/${tag}_shard_1rank_rccl_kernel_stats.csv && rm -rf $out/shard_prof1
for w in c4 c5 c2; do
  python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  python -c "
import json;d=json.load(open('$out/bench_$w.json'));print('$w', round(d['ms_per_step'],4), d['kernel_ms'], d['roofline'], d.get('also'))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_prof -- python3 bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_c4_under_rocprof.json 2> $out/c4_prof.err || { tail -5 $out/c4_prof.err; exit 1; }
cp "$(find $out/c4_prof -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_c4.csv && rm -rf $out/c4_prof
grep "spmf::" $out/${tag}_shard_kernel_stats.csv | cut -c1-50,200-330 | head -8
