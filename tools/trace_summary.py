"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / min / median / mean / max ns.

    python tools/trace_summary.py <dir-or-kernel_trace.csv> [substring-filter]
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("bitnet_hip::", "")
    return name[:60]


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    d = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if filt and filt not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["VGPR_Count"], r["LDS_Block_Size"])
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':60s} {'WGs':>6s} {'vgpr':>5s} {'lds':>7s} {'n':>6s} {'min':>8s} {'med':>8s} {'mean':>8s} {'max':>8s}  (ns)")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print(f"{k[0]:60s} {k[1]:6d} {k[2]:>5s} {k[3]:>7s} {len(v):6d} {v[0]:8d} {v[len(v)//2]:8d} {sum(v)//len(v):8d} {v[-1]:8d}")


if __name__ == "__main__":
    main()
