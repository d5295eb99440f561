set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_18.log
: > $L
for g in 0 -1 6 9 18 0; do
  echo "== wgroup $g" | tee -a $L
  BITNET_HIP_GEMM_WGROUP=$g timeout -k 10 200 python3 tools/ablate_gemm.py 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  BITNET_HIP_GEMM_WGROUP=$g timeout -k 10 200 python3 tools/ablate_f16a.py i2s 0 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for g in 0 -1 0 -1; do
  echo "== prefill wgroup $g" | tee -a $L
  BITNET_HIP_GEMM_WGROUP=$g timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
  BITNET_HIP_GEMM_WGROUP=$g timeout -k 10 200 python3 tools/perf_prefill_once.py i2s 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
done
