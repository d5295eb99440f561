set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_12.log
: > $L
BITNET_HIP_GEMM_FP6=1 timeout -k 10 200 python3 tools/ab_fp6.py --m 4096 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $L
BITNET_HIP_GEMM_FP6=1 timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
BITNET_HIP_GEMM_FP6=0 timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
BITNET_HIP_GEMM_FP6=1 timeout -k 10 200 python3 tools/perf_prefill_once.py qk256 3 30 2>&1 | grep -v amdgpu.ids | tee -a $L
BITNET_HIP_GEMM_FP6=1 timeout -k 10 300 bash tools/kstats_cmd.sh pf_fp6 tools/perf_prefill_once.py qk256 2 30 2>&1 | tee -a $L
