#!/bin/bash
# Round-N profile collection on the GPU box (one rocprofv3 run per counter group, as the pool requires):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r03'
# Output: gpurun_out/<tag>/{bench,C3,C4,C5}/{kt,pmc_*}; condense with tools/profile_summary.py.
set -o pipefail
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --no-cpu --no-secondary"
run() { d=$1; shift; timeout -k 10 240 rocprofv3 "$@" > "$OUT/$d.log" 2>&1; }
# headline command
python3 $REPO/bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
run bench_kt --kernel-trace --stats --output-format csv -d "$OUT/bench/kt" -- $B --steps 200 || exit 1
run bench_full --kernel-trace --stats --output-format csv -d "$OUT/bench/kt_full" -- python3 $REPO/bench.py --no-cpu --steps 200 || exit 1
run bench_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/bench/pmc_fetch" -- $B --steps 30 || exit 1
run bench_write --pmc WRITE_SIZE --output-format csv -d "$OUT/bench/pmc_write" -- $B --steps 30 || exit 1
run bench_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/bench/pmc_sq1" -- $B --steps 30 || exit 1
run bench_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d "$OUT/bench/pmc_sq2" -- $B --steps 30 || exit 1
run bench_sq3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/bench/pmc_sq3" -- $B --steps 30 || exit 1
run bench_sq4 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/bench/pmc_sq4" -- $B --steps 30 || true
run bench_sq5 --pmc SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/bench/pmc_sq5" -- $B --steps 30 || true
# secondary configurations: kernel trace + HBM traffic + MFMA busy
for C in C3 C4 C5; do
  P="python3 $REPO/tools/prof_config.py $C --steps 2"
  run ${C}_kt --kernel-trace --stats --output-format csv -d "$OUT/$C/kt" -- $P || exit 1
  run ${C}_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/$C/pmc_fetch" -- $P || exit 1
  run ${C}_write --pmc WRITE_SIZE --output-format csv -d "$OUT/$C/pmc_write" -- $P || exit 1
  run ${C}_sq3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$C/pmc_sq3" -- $P || exit 1
  echo "$C done"
done
# keep the merge-back small: counter CSVs of a 1M-point run are large only in dispatch count, agent info is not needed
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"
