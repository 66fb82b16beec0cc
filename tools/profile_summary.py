"""Condense rocprofv3 CSV output into the summaries committed under profiles/.

    python tools/profile_summary.py <rocprof_out_dir> <profiles/rNN_name>     # writes <name>.md and <name>.json

Expects sub-directories produced by separate runs of the SAME command (bench.py), as the pool requires:
  kt/         rocprofv3 --kernel-trace --stats --output-format csv
  pmc_fetch/  rocprofv3 --pmc FETCH_SIZE
  pmc_write/  rocprofv3 --pmc WRITE_SIZE
  pmc_sq*/    rocprofv3 --pmc SQ_* (any number of passes)
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B
per 128-B request for wide coalesced reads, so the read side is reported both raw and x2 (upper bound).
"""
import collections
import csv
import glob
import json
import os
import sys


def rows(pattern):
    out = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            out.extend(csv.DictReader(fh))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    summary = {"kernels": [], "pmc": {}}
    md = ["# rocprofv3 summary", "", f"source: `{src}` (one rocprofv3 run per counter group, same command)", ""]
    ks = rows(os.path.join(src, "kt", "**", "*kernel_stats.csv"))
    if ks:
        md += ["## kernel-trace --stats", "", "| kernel | calls | avg µs | min µs | max µs | % |", "|---|---|---|---|---|---|"]
        for r in sorted(ks, key=lambda r: -float(r["Percentage"]))[:8]:
            md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                      f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
            summary["kernels"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3})
        md.append("")
    kt = rows(os.path.join(src, "kt", "**", "*kernel_trace.csv"))
    jet = [r for r in kt if "jet_kernel" in r.get("Kernel_Name", "")]
    if jet:
        r = jet[-1]
        md += ["dominant kernel resources: " + ", ".join(f"{k}={r[k]}" for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r), ""]
    pmc = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for r in rows(os.path.join(d, "**", "*counter_collection.csv")):
            if "jet_kernel" in r["Kernel_Name"]:
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pmc:
        md += ["## PMC, dominant kernel (mean per dispatch)", "", "| counter | value |", "|---|---|"]
        for k in sorted(pmc):
            v = sum(pmc[k]) / len(pmc[k])
            summary["pmc"][k] = v
            md.append(f"| {k} | {v:.4g} |")
        md.append("")
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            f_kib, w_kib = summary["pmc"]["FETCH_SIZE"], summary["pmc"]["WRITE_SIZE"]
            summary["hbm_bytes_per_launch"] = {"read_raw": f_kib * 1024, "read_x2": 2 * f_kib * 1024, "write": w_kib * 1024}
            md += [f"HBM-side traffic per launch: read {f_kib * 1024 / 1e6:.1f} MB raw (≤ {2 * f_kib * 1024 / 1e6:.1f} MB with the gfx950 "
                   f"FETCH_SIZE x2 correction), write {w_kib * 1024 / 1e6:.1f} MB (tape slab + float atomics).", ""]
        if "SQ_WAVE_CYCLES" in pmc:
            wc = summary["pmc"]["SQ_WAVE_CYCLES"]
            parts = {k: summary["pmc"].get(k, 0.0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
            md += ["wave-cycle split: " + ", ".join(f"{k} {100 * v:.0f}%" for k, v in parts.items()), ""]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + ".md", "w").write("\n".join(md) + "\n")
    json.dump(summary, open(dst + ".json", "w"), indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    main()
