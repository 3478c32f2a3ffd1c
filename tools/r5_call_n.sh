#!/bin/bash
# round 5, call N: after the orderly p2p shutdown: the p2p / two-rank / launch tests, the workgroup sweep at world 2 and 4,
# the bench rehearsal over the peer kernel, and the counter traffic of the final sources again (api.hip changed: new stamp)
export TMPDIR=/tmp
o=gpurun_out/r5p
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_two_rank.py tests/test_gpu_bench_launch.py -q > $o/tests.log 2>&1
rc=$?; echo "tests rc=$rc" > $o/tests.rc; tail -3 $o/tests.log
[ $rc -lt 2 ] || exit 1
grep -E "^(FAILED|ERROR)" $o/tests.log
timeout -k 10 500 python tools/p2p_sweep.py > $o/p2p_sweep.jsonl 2> $o/p2p_sweep.err || { grep -v "Gloo\|Feature\|amdgpu\|socket" $o/p2p_sweep.err | tail -8; }
cat $o/p2p_sweep.jsonl
for w in 2 4; do
  SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 SPMF_BENCH_COMM=p2p timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w \
    --master-addr 127.0.0.1 --master-port $((29610 + w)) bench.py --gpus $w --steps 10 --warmup 2 --no-cpu-baseline --no-extras \
    > $o/bench_p2p_w$w.json 2> $o/bench_p2p_w$w.err
  rc=$?; [ $rc -eq 0 ] || { echo "world $w rc=$rc"; tail -8 $o/bench_p2p_w$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$o/bench_p2p_w$w.json')); print('world', $w, 'ms', round(d['ms_per_step'],4), d['collective'], 'elbo_x', d['elbo_x'])
"
done
bash tools/pmc.sh $o/pmc --workload c3 > $o/pmc_summary.txt 2> $o/pmc.err || { tail -5 $o/pmc.err; exit 1; }
python3 tools/pmc_traffic.py $o/pmc c3 $o/pmc_traffic.json
python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$o/bench.json')); print('C3', d['value'], d['ms_per_step'], d['roofline'], d['roofline_l2'])
"
