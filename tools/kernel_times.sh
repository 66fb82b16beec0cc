#!/bin/bash
# kernel-trace timing of one configuration's kernels: gpurun -- 'bash tools/kernel_times.sh C5 substring'
C=${1:-C5}; K=${2:-wres16}
REPO=$(pwd); OUT=$REPO/gpurun_out/w16exp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/x; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x -- python3 $REPO/tools/prof_config.py $C --steps 2 > $OUT/x.log 2>&1 || exit 1
grep -h "$K" $OUT/x/*/*kernel_stats.csv | cut -d, -f1-4 | tr -d '"'
