#!/bin/bash
# Developer tool (GPU box): counter passes over the f16 chain's gate|up launch (k_gemm_f16h + k_gemm_f16a), one rocprofv3 run per counter group: tools/pmc_f16a.sh [i2s|qk256]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
F=${1:-i2s}
O=gpurun_out/pmc_f16a_$F
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_MFMA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 tools/pmc_f16a_once.py $F > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; echo "group $i failed: $grp"; continue; }
done
python3 - $O <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: [0, 0])
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for fam in ("k_gemm_f16h", "k_gemm_f16a"):  # (round 5: the wide launches run k_gemm_f16h on their whole rounds + k_gemm_f16a on the rest)
            if fam in r["Kernel_Name"]:
                t = tot[(fam, r["Counter_Name"])]; t[0] += float(r["Counter_Value"]); t[1] += 1
for (fam, k), (v, n) in sorted(tot.items()):
    print(f"{fam:12s} {k:32s} {v / n:16.0f} per launch ({n} launches)")
PY
