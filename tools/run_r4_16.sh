set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_16.log
: > $L
BITNET_HIP_GEMM_RING=1 timeout -k 10 600 python3 -m pytest tests/test_bench_prefill_instance.py tests/test_f16_chain.py -x -q -k "f16 or prefill_2_digits" 2>&1 | tail -8 | tee -a $L
for r in 0 1 0 1; do
  echo "== ring $r" | tee -a $L
  BITNET_HIP_GEMM_RING=$r timeout -k 10 200 python3 tools/ablate_f16a.py i2s 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  BITNET_HIP_GEMM_RING=$r timeout -k 10 200 python3 tools/perf_prefill_once.py i2s 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
  BITNET_HIP_GEMM_RING=$r timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
done
