"""Summarise -Rpass-analysis=kernel-resource-usage logs: python tools/regs.py <log> [...]"""
import re
import subprocess
import sys

KEYS = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("spill", r"VGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
for path in sys.argv[1:]:
    txt = open(path).read()
    for b in txt.split("Function Name: ")[1:]:
        name = b.split()[0]
        vals = []
        for label, k in KEYS:
            m = re.search(k + r": (\d+)", b)
            vals.append(f"{label} {int(m.group(1)) if m else -1}")
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(pinn::[A-Za-z:]*Args\)", "", dn.replace("pinn::lm::", "").replace("pinn::", ""))
        print(f"{dn:60s} " + " ".join(vals))
