#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the bench command      -> gpurun_out/prof_<tag>/trace
#   2. a separate --pmc FETCH_SIZE pass (gfx950: FETCH_SIZE has 3 TCC slots, so it goes
#      alone; never combined with trace domains)       -> gpurun_out/prof_<tag>/pmc_fetch
#   3. a separate --pmc WRITE_SIZE pass                 -> gpurun_out/prof_<tag>/pmc_write
# (The PMC passes run tools/pmc_gemv.py: the same fused GEMV launches issued eagerly from
#  Python; rocprofv3 --pmc segfaults on this stack when the workload replays hipGraphs.)
# then summarises into gpurun_out/prof_<tag>/summary.{md,json}.  Copy what should be
# judged into profiles/ afterwards (gpurun_out/ is scratch).
#   usage: tools/profile_round.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}
shift
ARGS=${@:-"--steps 32 --warmup 4 --no-cpu-baseline"}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.log" 2>&1 || { tail -5 "$OUT/bench_trace.log"; exit 1; }
WL=c2; case "$ARGS" in *"--workload c3"*) WL=c3;; esac
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/pmc_gemv.py --workload $WL > "$OUT/pmc_fetch.log" 2>&1 || { grep -v "^    @" "$OUT/pmc_fetch.log" | tail -5; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/pmc_gemv.py --workload $WL > "$OUT/pmc_write.log" 2>&1 || { grep -v "^    @" "$OUT/pmc_write.log" | tail -5; exit 1; }
python3 tools/profile_summary.py "$OUT" "$TAG"
