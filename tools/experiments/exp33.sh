set -x
mkdir -p gpurun_out/e33
run() { # tag, args...
  tag=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/e33/$tag.json 2> gpurun_out/e33/$tag.err || tail -5 gpurun_out/e33/$tag.err
  python -c "
import json;d=json.load(open('gpurun_out/e33/$tag.json'));print('$tag', round(d['ms_per_step'],4), d['kernel_ms'], d['config'].get('rows'))"
}
for pr in 8192 8384 10432 11392 12544; do run c3_$pr --panel-rows $pr --steps 10 --warmup 3; done
for pr in 8192 7872 15680 5248 3968; do run shard_$pr --rows 125000 --panel-rows $pr --steps 50 --warmup 5; done
for pr in 8192 7872 5696; do run c4_$pr --workload c4 --panel-rows $pr --steps 5 --warmup 2; done
