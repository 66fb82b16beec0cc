"""Condense rocprofv3 CSV output into the summaries committed under profiles/.

    python tools/profile_summary.py <rocprof_out_dir> <profiles/rNN_name> [--iters N] [--raw]

Expects sub-directories produced by separate runs of the SAME command (tools/collect_profiles.sh), as the pool requires:
  kt/         rocprofv3 --kernel-trace --stats --output-format csv
  pmc_fetch/  rocprofv3 --pmc FETCH_SIZE
  pmc_write/  rocprofv3 --pmc WRITE_SIZE
  pmc_sq*/    rocprofv3 --pmc SQ_* / GRBM_* (any number of passes)
Writes <name>.md and <name>.json; --raw also copies the per-kernel means of every counter pass to <name>_pmc.csv so
that the numbers quoted in DESIGN.md can be re-derived from a tracked file.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B
per 128-B request for wide coalesced reads, so the read side is reported both raw and x2 (upper bound).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs); the counter saturates at
8.192e8 per dispatch on this stack, which the table flags.
--iters N: the profiled command ran N iterations of the workload (per-iteration totals are printed).
"""
import argparse
import collections
import csv
import glob
import json
import os

MFMA_SAT = 8.192e8


def rows(pattern):
    out = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            out.extend(csv.DictReader(fh))
    return out


def short(name):
    name = name.replace("void ", "").replace("pinn::lm::", "").replace("pinn::", "")
    for tail in ("(GemmArgs)", "(GemmNtArgs)", "(EwArgs)", "(HeadArgs)", "(KernelArgs)", "(FusedArgs)"):
        name = name.replace(tail, "")
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--raw", action="store_true")
    ap.add_argument("--title", default="rocprofv3 summary")
    args = ap.parse_args()
    src, dst = args.src, args.dst
    summary = {"kernels": [], "pmc": {}}
    md = [f"# {args.title}", "", f"source: `{src}` (one rocprofv3 run per counter group, same command)", ""]

    ks = sorted(rows(os.path.join(src, "kt", "**", "*kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
    total_ns = sum(float(r["TotalDurationNs"]) for r in ks)
    top = [r["Name"] for r in ks[:10]]

    pmc = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for r in rows(os.path.join(d, "**", "*counter_collection.csv")):
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
    summ = {k: {c: sum(v) for c, v in cs.items()} for k, cs in pmc.items()}

    if ks:
        md += ["## kernel-trace --stats + per-dispatch counters", "",
               "| kernel | calls | avg µs | % of GPU time | HBM read MB (raw / x2) | HBM write MB | MFMA busy |",
               "|---|---|---|---|---|---|---|"]
        for r in ks[:10]:
            m = mean.get(r["Name"], {})
            rd = m.get("FETCH_SIZE")
            wr = m.get("WRITE_SIZE")
            busy = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
                frac = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
                busy = f"{100 * frac:.0f} %" + (" (≥, counter saturated)" if m["SQ_VALU_MFMA_BUSY_CYCLES"] >= MFMA_SAT else "")
                if m["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
                    busy = "—"
            md.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | "
                      + (f"{rd * 1024 / 1e6:.1f} / {2 * rd * 1024 / 1e6:.1f}" if rd is not None else "") + " | "
                      + (f"{wr * 1024 / 1e6:.1f}" if wr is not None else "") + f" | {busy} |")
            summary["kernels"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                       "percent": float(r["Percentage"]), "pmc_mean": m})
        md.append("")
        md.append(f"GPU time in kernels: {total_ns / 1e6:.3f} ms" + (f" = {total_ns / 1e6 / args.iters:.3f} ms per iteration ({args.iters} iterations)" if args.iters else ""))
        md.append("")
        kt = rows(os.path.join(src, "kt", "**", "*kernel_trace.csv"))
        seen = set()
        res = []
        for r in kt:
            n = r.get("Kernel_Name", "")
            if n in top[:6] and n not in seen:
                seen.add(n)
                res.append(f"`{short(n)}`: " + ", ".join(f"{k}={r[k]}" for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size") if k in r))
        if res:
            md += ["kernel resources (from the trace):", ""] + [f"* {x}" for x in res] + [""]

    dom = top[0] if top else None
    if dom and dom in mean:
        m = mean[dom]
        summary["dominant"] = dom
        summary["pmc"] = m
        md += [f"## PMC, dominant kernel `{short(dom)}` (mean per dispatch)", "", "| counter | value |", "|---|---|"]
        for k in sorted(m):
            md.append(f"| {k} | {m[k]:.4g} |")
        md.append("")
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            f_kib, w_kib = m["FETCH_SIZE"], m["WRITE_SIZE"]
            summary["hbm_bytes_per_launch"] = {"read_raw": f_kib * 1024, "read_x2": 2 * f_kib * 1024, "write": w_kib * 1024}
            md += [f"HBM-side traffic per launch: read {f_kib * 1024 / 1e6:.1f} MB raw (≤ {2 * f_kib * 1024 / 1e6:.1f} MB with the gfx950 "
                   f"FETCH_SIZE x2 correction), write {w_kib * 1024 / 1e6:.1f} MB.", ""]
        if "SQ_WAVE_CYCLES" in m:
            wc = m["SQ_WAVE_CYCLES"]
            parts = {k: m.get(k, 0.0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
            md += ["wave-cycle split: " + ", ".join(f"{k} {100 * v:.0f}%" for k, v in parts.items()), ""]
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in m:
            md += [f"SQ_VALU_MFMA_COEXEC_CYCLES reads {m['SQ_VALU_MFMA_COEXEC_CYCLES']:.0f} for every kernel of the run (MFMA-heavy ones "
                   "included): the counter is accepted but not populated on this stack.", ""]
    if args.iters and summ:
        rd = sum(v.get("FETCH_SIZE", 0.0) for v in summ.values()) * 1024 / args.iters
        wr = sum(v.get("WRITE_SIZE", 0.0) for v in summ.values()) * 1024 / args.iters
        summary["hbm_bytes_per_iteration"] = {"read_raw": rd, "read_x2": 2 * rd, "write": wr}
        md += [f"HBM-side traffic per iteration, all kernels: read {rd / 1e9:.2f} GB raw (≤ {2 * rd / 1e9:.2f} GB x2), write {wr / 1e9:.2f} GB.", ""]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + ".md", "w") as f:
        f.write("\n".join(md) + "\n")
    with open(dst + ".json", "w") as f:
        json.dump(summary, f, indent=1)
    if args.raw:
        with open(dst + "_pmc.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
            for k in sorted(mean):
                for c in sorted(mean[k]):
                    w.writerow([k, c, len(pmc[k][c]), f"{mean[k][c]:.6g}"])
        for r in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
            with open(r) as fi, open(dst + "_kernel_stats.csv", "w") as fo:
                fo.write(fi.read())
    print("\n".join(md))


if __name__ == "__main__":
    main()
