#!/bin/bash
# Developer tool: cycles and clock of the matmul VARIANTS (tools/pmc_gemm_clock.sh's method, random operands only):
# does a variant that needs fewer cycles get them back as time, or does the chip lower its clock?
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/pmc_gemm_variants
rm -rf $O; mkdir -p $O
for v in base rw8v2 w1; do
  unset BITNET_HIP_GEMM_RW8 BITNET_HIP_GEMM_VAR BITNET_HIP_GEMM_W1
  case $v in
    rw8v2) export BITNET_HIP_GEMM_RW8=1 BITNET_HIP_GEMM_VAR=2 ;;
    w1) export BITNET_HIP_GEMM_W1=1 ;;
  esac
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/${v}_trace -- python3 tools/pmc_gemm_once.py 32768 random 6 > $O/${v}_trace.log 2>&1 || { tail -5 $O/${v}_trace.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/${v}_g1 -- python3 tools/pmc_gemm_once.py 32768 random 6 > $O/${v}_g1.log 2>&1 || tail -5 $O/${v}_g1.log
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${v}_g2 -- python3 tools/pmc_gemm_once.py 32768 random 6 > $O/${v}_g2.log 2>&1 || tail -5 $O/${v}_g2.log
done
python3 - <<'PY'
import csv, glob, collections
for v in ("base", "rw8v2", "w1"):
    dur = []
    for f in glob.glob(f"gpurun_out/pmc_gemm_variants/{v}_trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm_mfma" in r["Kernel_Name"] or "k_gemm_w1" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    dur = sorted(dur)[len(dur) // 2] if dur else float("nan")
    tot = collections.defaultdict(lambda: [0.0, 0])
    name = ""
    for f in glob.glob(f"gpurun_out/pmc_gemm_variants/{v}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm_mfma" in r["Kernel_Name"] or "k_gemm_w1" in r["Kernel_Name"]:
                t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1; name = r["Kernel_Name"][:60]
    c = {k: x / n for k, (x, n) in tot.items()}
    clk = c.get("GRBM_GUI_ACTIVE", float("nan")) / 8 / dur / 1e3
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (1024 * clk * 1e3 * dur)
    print(f"{v:6s} {name:60s} {dur:8.1f} us  cycles/XCD {c.get('GRBM_GUI_ACTIVE', 0) / 8 / 1e6:7.3f} M  clock {clk:.3f} GHz  matrix pipe busy {busy:.3f}")
PY
