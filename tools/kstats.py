"""Print the kernel_stats.csv of a rocprofv3 --kernel-trace --stats run as a short table: python tools/kstats.py <dir>"""
import csv
import glob
import os
import sys

for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{f}: total {tot / 1e6:.3f} ms")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print(f"  {float(r['Percentage']):6.2f}%  calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
