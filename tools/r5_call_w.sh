#!/bin/bash
# round 5, second session: bench.py's N > 1 path rehearsed on one GPU with the final library (ranks = processes on the
# one card, gloo for torch.distributed, the peer-pointer kernel as the step's collective) at world 2 and 4, and the auto pick
export TMPDIR=/tmp
o=gpurun_out/r5z
mkdir -p $o
for w in 2 4; do
  SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 SPMF_BENCH_COMM=p2p timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w \
    --master-addr 127.0.0.1 --master-port $((29710 + w)) bench.py --gpus $w --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
    > $o/bench_p2p_w$w.json 2> $o/bench_p2p_w$w.err
  rc=$?; [ $rc -eq 0 ] || { echo "world $w rc=$rc"; tail -8 $o/bench_p2p_w$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$o/bench_p2p_w$w.json')); print('world', $w, 'ms', round(d['ms_per_step'],4), d['config']['allreduce_transport'][:40], d['collective'], 'elbo_x', d['elbo_x'], 'nnf', d['n_nonfinite'])
"
done
SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29720 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
    > $o/bench_auto_w2.json 2> $o/bench_auto_w2.err
rc=$?; [ $rc -eq 0 ] || { echo "auto rc=$rc"; tail -8 $o/bench_auto_w2.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench_auto_w2.json')); print('auto world 2 ms', round(d['ms_per_step'],4), d['config']['allreduce_transport'][:60], d['collective'], 'elbo_x', d['elbo_x'])
"
