#!/usr/bin/env python3
"""Per (kernel, grid) medians from a rocprofv3 kernel trace: python tools/trace_shapes.py <dir or csv> [name filter ...]"""
import collections, csv, glob, statistics, sys
src = sys.argv[1]
f = src if src.endswith(".csv") else sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True))[-1]
flt = sys.argv[2:] or ["k_gemm", "k_quant", "k_prefill", "k_rows"]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(x in n for x in flt):
        d[(n.split("(")[0].replace("void bitnet_hip::", "").replace("bitnet_hip::", ""), r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items()):
    print(f"{k[0]:48s} grid {k[1]:>8s} x {k[2]:>4s}  calls {len(v):4d}  median {statistics.median(v):8.1f}  min {min(v):8.1f}  sum {sum(v) / 1e3:8.2f} ms")
    tot += sum(v)
print(f"total {tot / 1e3:.2f} ms")
