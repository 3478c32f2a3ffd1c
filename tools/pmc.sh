#!/bin/bash
# usage: tools/pmc.sh <outdir> <bench args...> ; separate PMC passes (no trace domains mixed in)
# then:  python tools/pmc_traffic.py <outdir> <workload> profiles/pmc_traffic.json
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err || { tail -5 $out/p$i.err; }
done
python3 - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'spmf' not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:24s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY
