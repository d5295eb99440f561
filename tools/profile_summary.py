"""Summarise one tools/profile_round.sh run: per-kernel time (rocprofv3 --kernel-trace) and
HBM traffic per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so fetch bytes =
2 * FETCH_SIZE * 1024; WRITE_SIZE reads exactly for 16-B-per-lane streaming stores
(bytes = WRITE_SIZE * 1024).  Other access widths are uncalibrated.

    python tools/profile_summary.py gpurun_out/prof_r02/c2 r02 c2

Writes <dir>/summary.{md,json} and, into profiles/: <tag>_<wl>_summary.{md,json}, <tag>_<wl>_kernel_stats.csv (the SAME rows),
<tag>_<wl>_bench.json (the un-profiled bench line of the same sitting) and traffic_<wl>.json for the workload's dominant I2_S
GEMV (what bench.py reports as roofline.traffic).  Files of other rounds for that workload are removed.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"(?:bitnet_hip::)?([A-Za-z0-9_]+(?:<[^(]*>)?)", name)
    if name.startswith("_ZN10bitnet_hip"):
        name = name.replace("12_GLOBAL__N_1", "")
        m2 = re.match(r"_ZN10bitnet_hip\d+([a-z_0-9A-Z]+?)(?:I|E)", name)
        return m2.group(1) if m2 else name[:40]
    return (m.group(1) if m else name)[:60]


DOMINANT = ("k_gemv_q<8, 5", "k_gemv_mfma<8, 5")  # the fused gate|up GEMV (RING 5: paired matrix at K = 2560)
DOMINANT_PREFILL = ("k_gemm_mfma", "k_gemm_f16a", "k_gemm_fp6")  # prefill passes: the tiled matmul with the largest total time (gate|up)


def newest(pattern_dir: str, suffix: str):
    """ONE sitting: the newest file below a pass directory.  gpurun merges every call's output into the local gpurun_out/
    without removing older files, so a directory can hold several sittings (one <pid>_*.csv each) -- VERDICT r02."""
    files = glob.glob(os.path.join(pattern_dir, "**", "*" + suffix), recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
    trace = newest(os.path.join(out_dir, "trace"), "kernel_trace.csv")
    k = collections.defaultdict(list)
    grids = collections.defaultdict(set)
    for f in trace:
        for r in csv.DictReader(open(f)):
            # workgroups of the whole (possibly 2-D / 3-D) grid: the counter files give the product, so the key uses it too
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            key = (short(r["Kernel_Name"]), grid // max(1, wg))
            k[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {}
    for cname, sub, mult in (("FETCH_SIZE", "pmc_fetch", 2.0 * 1024.0), ("WRITE_SIZE", "pmc_write", 1024.0)):
        d = collections.defaultdict(list)
        for f in newest(os.path.join(out_dir, sub), "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname:
                    continue
                wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 1
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, wg))
                d[key].append(float(r["Counter_Value"]) * mult)
        pmc[cname] = d
    rows = []
    total = sum(sum(v) for v in k.values())
    for key, v in sorted(k.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        fetch = pmc["FETCH_SIZE"].get(key)
        write = pmc["WRITE_SIZE"].get(key)
        rows.append({
            "kernel": key[0], "workgroups": key[1], "calls": len(v), "total_ms": round(sum(v) / 1e6, 3),
            "pct": round(100.0 * sum(v) / total, 2), "avg_us": round(sum(v) / len(v) / 1e3, 3), "min_us": round(v[0] / 1e3, 3),
            "med_us": round(v[len(v) // 2] / 1e3, 3), "max_us": round(v[-1] / 1e3, 3),
            "hbm_fetch_bytes_per_launch": round(sum(fetch) / len(fetch)) if fetch else None,
            "hbm_write_bytes_per_launch": round(sum(write) / len(write)) if write else None,
        })
    sitting = "unknown"
    try:
        sitting = open(os.path.join(os.path.dirname(os.path.abspath(out_dir)), "SITTING")).read().strip()
    except OSError:
        pass
    with open(os.path.join(out_dir, "summary.json"), "w") as f:
        json.dump({"tag": tag, "sitting": sitting, "trace_file": os.path.basename(trace[0]) if trace else None, "kernels": rows}, f, indent=1)
    with open(os.path.join(out_dir, "summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary {tag} {wl} (sitting {sitting}: one trace file, one file per counter pass)\n\n")
        f.write("Times: `rocprofv3 --kernel-trace --stats`; bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, ")
        f.write("fetch = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024.\n\n")
        f.write("| kernel | WGs | calls | total ms | % | avg us | min | med | max | HBM fetch B/launch | HBM write B/launch |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| {kernel} | {workgroups} | {calls} | {total_ms} | {pct} | {avg_us} | {min_us} | {med_us} | {max_us} | {hbm_fetch_bytes_per_launch} | {hbm_write_bytes_per_launch} |\n".format(**r))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    for old in glob.glob(os.path.join(prof, f"r[0-9][0-9]_{wl}_*")) + (glob.glob(os.path.join(prof, "r01_summary.*")) + glob.glob(os.path.join(prof, "r01_kernel_stats.csv")) + glob.glob(os.path.join(prof, "r01_bench_*.json")) if wl == "c2" else []):
        if not os.path.basename(old).startswith(tag + "_"):
            os.remove(old)
    import shutil

    shutil.copy(os.path.join(out_dir, "summary.md"), os.path.join(prof, f"{tag}_{wl}_summary.md"))
    shutil.copy(os.path.join(out_dir, "summary.json"), os.path.join(prof, f"{tag}_{wl}_summary.json"))
    if os.path.exists(os.path.join(out_dir, "perf_provider.log")):
        shutil.copy(os.path.join(out_dir, "perf_provider.log"), os.path.join(prof, f"{tag}_{wl}_host_wall.txt"))
    if os.path.exists(os.path.join(out_dir, "bench.json")):
        shutil.copy(os.path.join(out_dir, "bench.json"), os.path.join(prof, f"{tag}_{wl}_bench.json"))
    with open(os.path.join(prof, f"{tag}_{wl}_kernel_stats.csv"), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    dom = [r for r in rows if r["kernel"].startswith(DOMINANT_PREFILL if wl.startswith("prefill") else DOMINANT) and r["hbm_fetch_bytes_per_launch"] is not None]
    if dom:
        d = max(dom, key=lambda r: r["total_ms"])
        with open(os.path.join(prof, f"traffic_{wl}.json"), "w") as f:
            json.dump({"round": tag, "workload": wl, "kernel": d["kernel"], "workgroups": d["workgroups"],
                       "hbm_fetch_bytes_per_launch": d["hbm_fetch_bytes_per_launch"], "hbm_write_bytes_per_launch": d["hbm_write_bytes_per_launch"] or 0,
                       "avg_us_rocprof": d["avg_us"],
                       "sitting": sitting,
                       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate short passes over the same launch path as the timed run "
                                 "(tools/profile_round.sh lists the exact commands per workload); "
                                 "fetch = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024"}, f, indent=1)
    print(open(os.path.join(out_dir, "summary.md")).read()[:3000])


if __name__ == "__main__":
    main()
