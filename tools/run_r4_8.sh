set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 0 1; do
  echo "== BITNET_HIP_ATTN_HEAD_FAST=$v"
  BITNET_HIP_ATTN_HEAD_FAST=$v python3 tools/perf_prefill_once.py qk256 4 30 2>&1 | tail -1
done
for v in 0 1; do
export BITNET_HIP_ATTN_HEAD_FAST=$v
rm -rf gpurun_out/prof_attn$v; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_attn$v -- python3 tools/perf_prefill_once.py qk256 3 30 > gpurun_out/prof_attn$v.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('gpurun_out/prof_attn$v/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_prefill_attn' in r['Name']: print('head_fast=$v', r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
python -m pytest tests/test_prefill_parity.py -x -q -m gpu 2>&1 | tail -3
