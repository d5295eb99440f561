set -o pipefail
mkdir -p gpurun_out
tools/probes/mfma_power_probe > gpurun_out/mfma_power_probe.txt 2>&1
python -m pytest tests/test_gemm_parity.py tests/test_prefill_parity.py tests/test_bench_prefill_instance.py tests/test_dist_device.py -x -q -m gpu > gpurun_out/r4_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t1.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?" >> gpurun_out/r4_t1.log
BITNET_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 32 --c5-prompt 2048 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err; echo "bench n2 rc $?" >> gpurun_out/r4_t1.log
cat gpurun_out/mfma_power_probe.txt; tail -15 gpurun_out/r4_t1.log
