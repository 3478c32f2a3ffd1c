#!/bin/bash
# usage: tools/pmc_mfma.sh <outdir> <bench args...> ; matrix-core utilisation counters of the dense kernels
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/$out/p$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err || { tail -5 $out/p$i.err; }
done
python3 - "/tmp/$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'expdot' not in k and 'sigdot' not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:32s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY
