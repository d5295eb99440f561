set -o pipefail
mkdir -p gpurun_out
bash tools/profile_round.sh r04 > gpurun_out/profile_r04_c.log 2>&1; echo "profile rc $?" >> gpurun_out/profile_r04_c.log
grep "== \|rc" gpurun_out/profile_r04_c.log | tail -12
