#!/bin/bash
# FORM=int8|fp6 (fp6: run under BITNET_HIP_GEMM_FP6=1) variant of pmc_gemm_clock.sh, one output directory per form.
# Developer tool (GPU box): what clock does the chip hold inside the prefill matmul, and how busy is the matrix pipe at THAT clock?
# A long dispatch (32768 token rows: ~2.5 ms per launch; GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches under ~0.3 ms,
# MI355X_MICROARCH.md DVFS give-back) on random and on all-zero operands, one rocprofv3 pass per counter group + one trace pass.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/pmc_gemm_clock_${FORM:-int8}
rm -rf $O; mkdir -p $O
for data in random; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/${data}_trace -- python3 tools/pmc_gemm_once.py 32768 $data 6 > $O/${data}_trace.log 2>&1 || { tail -5 $O/${data}_trace.log; exit 1; }
  i=0
  for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/${data}_g$i -- python3 tools/pmc_gemm_once.py 32768 $data 6 > $O/${data}_g$i.log 2>&1 || { tail -5 $O/${data}_g$i.log; echo "group failed: $grp"; }
  done
done
python3 - ${FORM:-int8} <<'PY'
import csv, glob, collections, sys
FORMX = sys.argv[1]
for data in ("random",):
    dur = []
    for f in glob.glob(f"gpurun_out/pmc_gemm_clock_{FORMX}/{data}_trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in ("k_gemm_mfma", "k_gemm_w1", "k_gemm_fp6")):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    dur = sorted(dur)[len(dur) // 2] if dur else float("nan")
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc_gemm_clock_{FORMX}/{data}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in ("k_gemm_mfma", "k_gemm_w1", "k_gemm_fp6")):
                t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
    c = {k: v / n for k, (v, n) in tot.items()}
    print(f"== {data}: median kernel duration {dur:.1f} us (32768 tokens x 13824 rows x 2560 cols, 2 digits)")
    for k, v in sorted(c.items()):
        print(f"   {k:28s} {v:18.0f} per launch")
    if "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3  # GHz
        print(f"   effective clock = GRBM_GUI_ACTIVE / 8 / duration = {clk:.3f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk * 1e3 * dur):.3f}")
PY
