"""What PINN_FLAG_DETERMINISTIC costs on the headline configuration (it routes the call to the layer-major engine).
    python tools/det_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from pinnrl_amd import engine as E  # noqa: E402

for tag in ("C2", "C3"):
    name, net, eq, n = B.CONFIGS[tag]()
    torch.manual_seed(1)
    x, t = eq.generate_collocation_points(n, strategy="uniform")
    for det in (False, True):
        net.set_deterministic(det)
        prog, pd = net.program(), eq._pde_desc()
        flat = E.new_flat_grad(prog, B.dev)
        for _ in range(3):
            E.residual_loss_grad(prog, pd, x, t, 1.0 / x.shape[0], flat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            E.residual_loss_grad(prog, pd, x, t, 1.0 / x.shape[0], flat)
        torch.cuda.synchronize()
        print(f"{tag} {'deterministic' if det else 'default      '} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms")
