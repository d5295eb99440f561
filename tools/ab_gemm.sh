#!/bin/bash
# Developer A/B of the prefill matmul variants (env switches of kernels_gemm.hip), one process per variant:
#   bash tools/ab_gemm.sh  ->  gpurun_out/ab_gemm.log
mkdir -p gpurun_out
out=gpurun_out/ab_gemm.log
: > $out
for fmt in qk256 i2s; do
for cfg in "F16A=0" "F16A=1" "F16A=0" "F16A=1"; do
  echo "== $fmt $cfg" | tee -a $out
  env BITNET_HIP_GEMM_${cfg} python3 tools/perf_gemm.py --fmt $fmt --digits 2 --check --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $out
done
done
