#!/bin/bash
# Developer tool (GPU box): HBM / fabric fetch bytes per launch of the prompt's matmuls with and without the weight-stationary walk
# (BITNET_HIP_GEMM_WGROUP): one rocprofv3 --pmc FETCH_SIZE pass per setting over tools/perf_prefill_once.py <fmt> 1 4.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
F=${1:-qk256}
for g in ${GROUPS_LIST:-0 -1}; do
  O=gpurun_out/pmc_wgroup_${F}_$g
  rm -rf $O; mkdir -p $O
  export BITNET_HIP_GEMM_WGROUP=$g
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python3 tools/perf_prefill_once.py $F 1 4 > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  python3 - $O $g <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            key = (r["Kernel_Name"].split("(")[0].replace("void bitnet_hip::", ""), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
            t = tot[key]; t[0] += float(r["Counter_Value"]); t[1] += 1
print(f"== BITNET_HIP_GEMM_WGROUP={sys.argv[2]}")
for k, (v, n) in sorted(tot.items()):
    print(f"  {k[0]:36s} grid {k[1]:>9s}: fetch {2 * v * 1024 / n / 1e6:8.1f} MB per launch ({n} launches; 2 x FETCH_SIZE x 1024)")
PY
done
