#!/bin/bash
# Per-kernel stats of any python command (GPU box): tools/kstats_cmd.sh <tag> <script> [args]; prints the top kernels.
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT <<'P'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:7.2f}  {r['Percentage']}%")
P
