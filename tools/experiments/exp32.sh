set -x
mkdir -p gpurun_out/e32
run() { # tag, args...
  tag=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/e32/$tag.json 2> gpurun_out/e32/$tag.err || tail -5 gpurun_out/e32/$tag.err
  python -c "
import json;d=json.load(open('gpurun_out/e32/$tag.json'));print('$tag', round(d['ms_per_step'],4), d['kernel_ms'])"
}
for pr in 10240 12288 14336; do run c3_$pr --panel-rows $pr --steps 10 --warmup 3; done
for pr in 8192 12288 24576; do run shard_$pr --rows 125000 --panel-rows $pr --steps 50 --warmup 5; done
for pr in 6144 8192 12288; do run c4_$pr --workload c4 --panel-rows $pr --steps 5 --warmup 2; done
for pr in 8192 12288; do run c5_$pr --workload c5 --panel-rows $pr --steps 10 --warmup 3; done
for pr in 8192 12288; do run c2_$pr --workload c2 --panel-rows $pr --steps 50 --warmup 5; done
