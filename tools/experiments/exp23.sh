set -x
mkdir -p gpurun_out/e23
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bernoulli.py tests/test_gpu_mixed.py tests/test_gpu_rule_and_surface.py -m gpu -x -q > gpurun_out/e23/pytest.log 2>&1 || { tail -30 gpurun_out/e23/pytest.log; exit 1; }
tail -2 gpurun_out/e23/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/e23/prof -o e23 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/e23/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/e23/bench.err || tail -5 $GRAFT_REPO_ROOT/gpurun_out/e23/bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
d=json.load(open('gpurun_out/e23/bench.json'));print(round(d['ms_per_step'],4), d['kernel_ms'], {k:v for k,v in d.get('also',{}).items() if 'shard' in k})
f=glob.glob('gpurun_out/e23/prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
