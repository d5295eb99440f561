#!/bin/bash
# Developer A/B of the prefill matmul variants (env switches of kernels_gemm.hip), one process per variant:
#   bash tools/ab_gemm.sh  ->  gpurun_out/ab_gemm.log
mkdir -p gpurun_out
out=gpurun_out/ab_gemm.log
: > $out
for data in random zeros zero_x const_w; do
  echo "== data $data" | tee -a $out
  python3 tools/perf_gemm.py --digits 2 --reps 30 --data $data --shapes gate_up down 2>&1 | grep -v amdgpu.ids | tee -a $out
done
