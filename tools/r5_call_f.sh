#!/bin/bash
# round 5, call F: the bench's N > 1 path over the peer-pointer kernel, rehearsed on the one GPU (ranks = processes on
# the card, gloo for torch.distributed, SPMF_BENCH_COMM=p2p): world 2 and 4; the p2p / launch / two-rank tests
export TMPDIR=/tmp
o=gpurun_out/r5h
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_bench_launch.py tests/test_gpu_two_rank.py -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
for w in 2 4; do
  SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 SPMF_BENCH_COMM=p2p timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w \
    --master-addr 127.0.0.1 --master-port $((29510 + w)) bench.py --gpus $w --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
    > $o/bench_p2p_w$w.json 2> $o/bench_p2p_w$w.err
  rc=$?; [ $rc -eq 0 ] || { echo "world $w rc=$rc"; tail -8 $o/bench_p2p_w$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$o/bench_p2p_w$w.json')); print('world', $w, 'ms', round(d['ms_per_step'],4), d['config']['allreduce_transport'][:60], d['collective'], 'elbo_x', d['elbo_x'])
"
done
SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 SPMF_BENCH_COMM=torch timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
    > $o/bench_gloo_w2.json 2> $o/bench_gloo_w2.err || { tail -5 $o/bench_gloo_w2.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench_gloo_w2.json')); print('world 2 host-staged gloo: ms', round(d['ms_per_step'],4), 'elbo_x', d['elbo_x'])
"
