set -o pipefail
mkdir -p gpurun_out
bash tools/profile_round.sh r04 c2 stream prefill > gpurun_out/profile_r04_a.log 2>&1; echo "profile rc $?" >> gpurun_out/profile_r04_a.log
tail -5 gpurun_out/profile_r04_a.log
