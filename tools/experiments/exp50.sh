set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/e50
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e50/prof -- python3 tools/fit_c3.py 100 1 > gpurun_out/e50/fit.json 2> gpurun_out/e50/fit.err || { tail -5 gpurun_out/e50/fit.err; exit 1; }
f=$(find gpurun_out/e50/prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/e50/fit_kernel_stats.csv; rm -rf gpurun_out/e50/prof
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/e50/fit_kernel_stats.csv')):
    if 'spmf::' in r['Name'] and int(r['Calls']) >= 100:
        print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
cat gpurun_out/e50/fit.json
