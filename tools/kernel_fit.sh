#!/bin/bash
# Per-kernel fixed cost: kernel-trace of one configuration at several point counts, avg us per kernel and N.
#   gpurun -- 'bash tools/kernel_fit.sh C3 "50000 100000 200000"'
C=${1:-C3}; NS=${2:-"50000 100000 200000"}
REPO=$(pwd); OUT=$REPO/gpurun_out/kfit; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in $NS; do
  rm -rf $OUT/n$N
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$N -- python3 $REPO/tools/prof_config.py $C --steps 2 --points $N > $OUT/n$N.log 2>&1 || exit 1
done
python3 - $OUT $NS <<'PY'
import csv, glob, sys, collections
out, ns = sys.argv[1], [int(v) for v in sys.argv[2:]]
t = collections.defaultdict(dict)
for n in ns:
    for f in glob.glob(f"{out}/n{n}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lm_" in r["Name"] or "jet_" in r["Name"]:
                t[r["Name"][:64]][n] = float(r["AverageNs"]) / 1e3
print("kernel | " + " | ".join(f"N={n}" for n in ns) + " | fixed us (2-point fit, ends)")
for k, v in sorted(t.items(), key=lambda kv: -max(kv[1].values())):
    if len(v) == len(ns):
        a, b = ns[0], ns[-1]
        slope = (v[b] - v[a]) / (b - a)
        print(k, "|", " | ".join(f"{v[n]:.1f}" for n in ns), "|", f"{v[a] - slope * a:.1f}")
PY
