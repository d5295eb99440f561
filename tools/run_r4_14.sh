set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_14.log
: > $L
for h in 0 1 0 1; do
  echo "== hybrid $h" | tee -a $L
  BITNET_HOST_PREFILL_HYBRID=$h timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
done
BITNET_HOST_PREFILL_HYBRID=1 timeout -k 10 300 bash tools/kstats_cmd.sh pf_hyb tools/perf_prefill_once.py qk256 2 30 > /dev/null 2>&1
python3 tools/trace_shapes.py gpurun_out/ks_pf_hyb | tee -a $L
timeout -k 10 200 python3 tools/perf_prefill_once.py i2s 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
