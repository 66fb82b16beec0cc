"""Fixed cost of the fused launch: kernel time against tiles per workgroup (N = 8192 k points = k tiles on each of 256
CUs) for the headline network; the intercept of the line is what a launch pays beyond its tile rounds.
    python tools/fixed_overhead.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import _burgers  # noqa: E402
from pinnrl_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")
cfg, model, pde = _burgers(dev, hidden=128, layers=4, mapping=32, scale=10.0)
prog, pd = model.program(), pde._pde_desc()
flat = E.new_flat_grad(prog, dev)
rows = []
for k in (1, 2, 3, 4, 6, 8, 12):
    n = 8192 * k
    x = torch.rand(n, 1, device=dev) * 2 - 1
    t = torch.rand(n, 1, device=dev)
    for _ in range(30):
        E.residual_loss_grad(prog, pd, x, t, 1.0 / n, flat)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in ev:
        a.record()
        E.residual_loss_grad(prog, pd, x, t, 1.0 / n, flat)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    rows.append((k, ms))
    print(f"k = {k:2d} tiles per CU: {1e3 * ms:8.1f} us")
k0, m0 = rows[0]
k1, m1 = rows[-1]
slope = (m1 - m0) / (k1 - k0)
print(f"per tile round {1e3 * slope:.1f} us, intercept {1e3 * (m0 - slope * k0):.1f} us")
