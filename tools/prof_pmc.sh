#!/bin/bash
# Stall / LDS counters of one secondary configuration:  gpurun -- 'bash tools/prof_pmc.sh C5 tag kernel_substring'
set -o pipefail
C=${1:-C5}; TAG=${2:-pmc}; KERN=${3:-lm_gemm}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/prof_config.py $C --steps 1"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/$C/pmc_sq1" -- $P > "$OUT/${C}_sq1.log" 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d "$OUT/$C/pmc_sq4" -- $P > "$OUT/${C}_sq4.log" 2>&1 || exit 1
find "$OUT" -name '*agent_info.csv' -delete
python3 - "$OUT/$C" "$KERN" <<'PY'
import csv, glob, sys, collections
src, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
