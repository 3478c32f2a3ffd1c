set -x
mkdir -p gpurun_out/e7
echo "== f32" > gpurun_out/e7/b3_err.txt; python tools/b3_err.py >> gpurun_out/e7/b3_err.txt 2>gpurun_out/e7/err0.log
for v in main b3p8 b3split b3p8split; do
  if [ $v = main ]; then unset SPMF_LIB_PATH; else export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so; fi
  echo "== bf16x3 $v" >> gpurun_out/e7/b3_err.txt
  SPMF_DENSE_BF16X3=1 python tools/b3_err.py >> gpurun_out/e7/b3_err.txt 2>gpurun_out/e7/err_$v.log
done
cat gpurun_out/e7/b3_err.txt
for v in b3p8 b3split b3p8split; do
  export SPMF_LIB_PATH=$PWD/spmf_amd/variants/libspmf_$v.so
  SPMF_DENSE_BF16X3=1 python bench.py --workload c4 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/e7/c4_$v.json 2> gpurun_out/e7/c4_$v.err || tail -5 gpurun_out/e7/c4_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/e7/c4_$v.json'));print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
done
