#!/bin/bash
# Regenerates EVERY file under profiles/ for one round from ONE sitting on the GPU box (run through gpurun):
#   for each workload (c2 c3 c4):
#     1. rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W ...`   -> <OUT>/<W>/trace
#     2. rocprofv3 --pmc FETCH_SIZE of the SAME bench.py launch path (hipGraph)     -> <OUT>/<W>/pmc_fetch
#     3. rocprofv3 --pmc WRITE_SIZE likewise (FETCH_SIZE takes 3 TCC slots: alone) -> <OUT>/<W>/pmc_write
#   then tools/profile_summary.py writes <tag>_<W>_summary.{md,json}, <tag>_<W>_kernel_stats.csv (the same numbers) and
#   traffic_<W>.json (what bench.py quotes as roofline.traffic), all into profiles/, and removes older rounds' files.
# Counter passes use a SHORT run (--prompt 8 --steps 8): rocprofv3's counter collection segfaults in the host process
# once a process has issued a few tens of thousands of dispatches (r01's "bench_pmc_fetch.log": reproduced in r02 with
# and without hipGraphs, eager or not -- 5.7k dispatches pass, the 30k-dispatch default bench does not); the launch
# path, kernels, shapes and weights are the timed run's.
#   usage: tools/profile_round.sh <tag>        (the program sits directly behind `--`: no env / bash -c hop under rocprofv3)
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"
mkdir -p "$OUT"
for WL in c2 c3 c4; do
    D="$OUT/$WL"
    mkdir -p "$D"
    TARGS="--workload $WL --steps 64 --warmup 4 --no-cpu-baseline"
    PARGS="--workload $WL --prompt 8 --steps 8 --warmup 0 --no-cpu-baseline"
    if [ "$WL" = c4 ]; then TARGS="--workload c4 --steps 64 --warmup 4 --no-cpu-baseline"; PARGS="--workload c4 --steps 8 --warmup 0 --no-cpu-baseline"; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 bench.py $TARGS > "$D/bench_trace.log" 2>&1 || { tail -5 "$D/bench_trace.log"; exit 1; }
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -- python3 bench.py $PARGS > "$D/pmc_fetch.log" 2>&1 || { grep -v "^    @" "$D/pmc_fetch.log" | tail -5; exit 1; }
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -- python3 bench.py $PARGS > "$D/pmc_write.log" 2>&1 || { grep -v "^    @" "$D/pmc_write.log" | tail -5; exit 1; }
    python3 bench.py --workload $WL --steps 128 --warmup 8 > "$D/bench.json" 2> "$D/bench.err" || { tail -5 "$D/bench.err"; exit 1; }
    python3 tools/profile_summary.py "$D" "$TAG" "$WL" || exit 1
    echo "== $WL done"
done
# the provider-trait ops (a9 / a10 / a12) and the a8 composite: device times per kernel
mkdir -p "$OUT/provider"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/provider/trace" -- python3 tools/perf_provider.py > "$OUT/provider/perf_provider.log" 2>&1 || { tail -5 "$OUT/provider/perf_provider.log"; exit 1; }
python3 tools/profile_summary.py "$OUT/provider" "$TAG" provider || exit 1
