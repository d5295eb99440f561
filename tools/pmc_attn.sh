#!/bin/bash
# Developer tool (GPU box): counter passes over the prefill attention kernel, one rocprofv3 run per counter group.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 tools/pmc_attn_once.py > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; echo "group $i failed: $grp"; continue; }
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0])
for f in glob.glob("gpurun_out/pmc_attn/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_prefill_attn" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k, (v, n) in sorted(tot.items()):
    print(f"{k:32s} {v / n:16.0f} per launch ({n} launches)")
PY
