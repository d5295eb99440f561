#!/bin/bash
# Regenerates EVERY file under profiles/ for one round from ONE sitting on the GPU box (run through gpurun):
#   for each decode workload (c2 c3 c4):
#     1. rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W ...`   -> <OUT>/<W>/trace
#     2. rocprofv3 --pmc FETCH_SIZE of the SAME bench.py launch path (hipGraph)     -> <OUT>/<W>/pmc_fetch
#     3. rocprofv3 --pmc WRITE_SIZE likewise (FETCH_SIZE takes 3 TCC slots: alone) -> <OUT>/<W>/pmc_write
#   the same three passes for the streaming-rate probe (`bench.py --workload stream`, both formats) and for ONE 4096-token
#   prefill per format (tools/perf_prefill_once.py: a few hundred dispatches, so the counter passes cover the prefill kernels),
#   then tools/profile_summary.py writes <tag>_<W>_summary.{md,json}, <tag>_<W>_kernel_stats.csv (the same numbers) and
#   traffic_<W>.json (what bench.py quotes as roofline.traffic), all into profiles/, and removes older rounds' files.
# ONE SITTING: the output directory carries the sitting's start time, and profile_summary.py reads only the NEWEST trace /
# counter file below each pass directory -- VERDICT r02: `gpurun` merges a call's gpurun_out/ into the local one WITHOUT removing
# what earlier calls left there, so a summary made from the merged directory mixed up to six sittings of different code.
# Counter passes use a SHORT run (--prompt 8 --steps 8): rocprofv3's counter collection segfaults in the host process
# once a process has issued a few tens of thousands of dispatches.  The saved backtrace (round 1's bench_pmc_fetch.log) ends
# in 15 frames of the HIP runtime / profiler libraries BELOW hipLaunchKernel -- this library's last frame is the kernel's
# host-side launch stub, which only forwards by-value arguments -- at a page-aligned fault address (a write running off the
# end of a mapping): not in this repository's code.  On a failed pass the whole log is kept, stack frames included.
#   usage: tools/profile_round.sh <tag> [passes]   passes: any of c2 c3 c4 stream prefill provider (default: all)
# The program sits directly behind `--`: no env / bash -c hop under rocprofv3.
set -o pipefail
TAG=${1:-r05}
shift
PASSES=${*:-stream prefill c2 c3 c4 provider}   # stream and prefill first: their traffic_*.json (written into profiles/ as each pass ends) are what the bench lines of c2 / c4 quote
export TMPDIR=/tmp
SIT=$(date +%Y%m%d_%H%M%S)
OUT=gpurun_out/prof_${TAG}_${SIT}
mkdir -p "$OUT"
echo "$SIT" > "$OUT/SITTING"
three_passes() {  # $1 = dir, $2.. = program + args for the trace pass; PMC_ARGS = args of the counter passes
    local D=$1; shift
    mkdir -p "$D"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- "$@" $TRACE_ARGS > "$D/trace.log" 2>&1 || { tail -5 "$D/trace.log"; return 1; }
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -- "$@" $PMC_ARGS > "$D/pmc_fetch.log" 2>&1 || { tail -40 "$D/pmc_fetch.log"; return 1; }
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -- "$@" $PMC_ARGS > "$D/pmc_write.log" 2>&1 || { tail -40 "$D/pmc_write.log"; return 1; }
}
for WL in $PASSES; do
    D="$OUT/$WL"
    case $WL in
    c2|c3|c4)
        TRACE_ARGS="--workload $WL --steps 64 --warmup 4 --no-cpu-baseline --no-stream --no-exact-check --no-also"
        PMC_ARGS="--workload $WL --prompt 8 --steps 8 --warmup 0 --no-cpu-baseline --no-stream --no-exact-check --no-also"
        if [ "$WL" = c4 ]; then PMC_ARGS="--workload c4 --steps 8 --warmup 0 --no-cpu-baseline --no-stream --no-exact-check --layers 4"; fi
        three_passes "$D" python3 bench.py || exit 1
        STEPS=128; if [ "$WL" = c4 ]; then STEPS=512; fi   # configs[3] as written: 4096-token prefill + 512 decode
        python3 bench.py --workload $WL --steps $STEPS --warmup 8 > "$D/bench.json" 2> "$D/bench.err" || { tail -5 "$D/bench.err"; exit 1; }
        python3 tools/profile_summary.py "$D" "$TAG" "$WL" || exit 1 ;;
    stream)
        for F in i2s qk256; do
            TRACE_ARGS="--workload stream --stream-format $F --stream-isolated"   # isolated launches: the per-kernel duration bench.py's i2s_stream.us_per_launch reports
            PMC_ARGS="--workload stream --stream-format $F --stream-launches 6"
            three_passes "${D}_$F" python3 bench.py || exit 1
            python3 tools/profile_summary.py "${D}_$F" "$TAG" "stream_$F" || exit 1
        done ;;
    prefill)
        for F in qk256 i2s; do
            TRACE_ARGS="$F"
            PMC_ARGS="$F 1 4"   # one repetition, 4 layers: the same launches per layer, a fifth of the dispatches
            three_passes "${D}_$F" python3 tools/perf_prefill_once.py || exit 1
            python3 tools/profile_summary.py "${D}_$F" "$TAG" "prefill_$F" || exit 1
        done ;;
    provider)
        # the provider-trait ops (a9 / a10 / a12) and the a8 composite: device times per kernel
        mkdir -p "$D"
        rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 tools/perf_provider.py > "$D/perf_provider.log" 2>&1 || { tail -5 "$D/perf_provider.log"; exit 1; }
        python3 tools/profile_summary.py "$D" "$TAG" provider || exit 1 ;;
    esac
    echo "== $WL done"
done
