#!/bin/bash
# One secondary configuration's kernel trace + MFMA-busy pass:  gpurun -- 'bash tools/prof_one.sh C5 tag'
set -o pipefail
C=${1:-C5}; TAG=${2:-one}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/prof_config.py $C --steps 2"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$C/kt" -- $P > "$OUT/${C}_kt.log" 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$C/pmc_sq3" -- $P > "$OUT/${C}_sq3.log" 2>&1 || exit 1
find "$OUT" -name '*agent_info.csv' -delete
python3 $REPO/tools/profile_summary.py "$OUT/$C" "$OUT/$C" --iters 3 > /dev/null 2>&1 || true
head -30 "$OUT/$C.md"
