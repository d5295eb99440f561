#!/bin/bash
# Three back-to-back bench runs of one workload (A/B comparisons: run-to-run noise is ~0.2 %).
#   usage (on the GPU box): bash tools/ab_bench.sh c3
W=${1:-c3}
for i in 1 2 3; do timeout -k 10 300 python bench.py --workload $W --steps 256 --warmup 8 --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['per_kernel']['attention']['us_per_launch'])"; done
