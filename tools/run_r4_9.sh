for m in 2304 3072 3584 4096 4608; do python3 tools/perf_gemm.py --digits 2 --reps 30 --shapes down o --m $m 2>&1 | grep -v amdgpu.ids | cut -c1-110; done
