set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_full_gpu.log
tail -5 gpurun_out/r4_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/run_sweeps.sh > gpurun_out/r4_sweeps.log 2>&1; echo "sweeps rc $?" >> gpurun_out/r4_sweeps.log
tail -12 gpurun_out/r4_sweeps.log
