# GPU box, round 4 (b): shard profiles (plain / one-rank RCCL), C5 / C4 / C2 bench lines and kernel stats, 200-epoch fit
set -x
export TMPDIR=/tmp
tag=r04
out=gpurun_out/ev_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_prof -- python3 bench.py --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/shard_bench.json 2> $out/shard_prof.err || { tail -5 $out/shard_prof.err; exit 1; }
cp "$(find $out/shard_prof -name '*kernel_stats.csv' | head -1)" $out/${tag}_shard_kernel_stats.csv && rm -rf $out/shard_prof
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_prof1 -- python3 bench.py --gpus 1 --rows 125000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/shard_1rank_bench.json 2> $out/shard_prof1.err || { tail -5 $out/shard_prof1.err; exit 1; }
cp "$(find $out/shard_prof1 -name '*kernel_stats.csv' | head -1)" $out/${tag}_shard_1rank_rccl_kernel_stats.csv && rm -rf $out/shard_prof1
for w in c5 c2 c4; do
  python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
done
for w in c5 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${w}_prof -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_${w}_under_rocprof.json 2> $out/${w}_prof.err || { tail -5 $out/${w}_prof.err; exit 1; }
  cp "$(find $out/${w}_prof -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_$w.csv && rm -rf $out/${w}_prof
done
SPMF_DENSE_BF16X3=0 python bench.py --workload c5 --no-extras --no-cpu-baseline > $out/bench_c5_f32mfma.json 2> $out/bench_c5_f32.err
python tools/fit_c3.py > $out/fit_c3.json 2> $out/fit_c3.err || tail -3 $out/fit_c3.err
echo evidence-b-done
