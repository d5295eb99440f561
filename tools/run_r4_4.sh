set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_f16_chain.py -x -q -m gpu > gpurun_out/r4_t4.log 2>&1; echo "pytest chain rc $?" >> gpurun_out/r4_t4.log
rm -rf gpurun_out/prof_chain; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain -- python3 tools/perf_prefill_once.py i2s 3 30 > gpurun_out/prof_chain.log 2>&1
tail -12 gpurun_out/r4_t4.log; tail -2 gpurun_out/prof_chain.log
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_chain/**/*kernel_stats.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(12), r['AverageNs'][:9].rjust(10), r['Percentage'][:6])
PY
