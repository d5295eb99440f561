set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_11.log
: > $L
timeout -k 10 300 python3 -m pytest tests/test_f16_chain.py -x -q 2>&1 | tail -5 | tee -a $L
timeout -k 10 200 python3 tools/ab_fp6.py --m 4096 2>&1 | grep -v amdgpu.ids | tee -a $L
for f in i2s qk256; do
  timeout -k 10 200 python3 tools/perf_prefill_once.py $f 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
done
BITNET_HIP_GEMM_FP6=0 timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
