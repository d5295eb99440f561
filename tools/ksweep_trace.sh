#!/bin/bash
# Developer tool (GPU box): the K sweep of tools/perf_gemm_ksweep.py under rocprofv3 --kernel-trace, so that the matmul kernel's own duration
# (without its row quantiser) can be read per K: tools/ksweep_trace.sh [flags = 16]; prints median duration per (kernel, K group).
export TMPDIR=/tmp
FL=${1:-16}
O=gpurun_out/ksweep_$FL
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/perf_gemm_ksweep.py $FL > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
grep -v amdgpu.ids $O/log.txt
python3 - $O <<'P'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_gemm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
groups, cur = [], []
for r in rows:
    key = (r['Kernel_Name'][:48], r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else '', r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''))
    if cur and (cur[0][0] != key or len(cur) == 23):
        groups.append(cur); cur = []
    cur.append((key, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
if cur: groups.append(cur)
for g in groups:
    d = sorted(x[1] for x in g)
    print(f"{g[0][0][0]:48s} grid {g[0][0][2]}x{g[0][0][3]} n={len(g):2d} median {statistics.median(d):8.2f} us  min {d[0]:8.2f}")
P
