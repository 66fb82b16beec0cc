"""One BASELINE configuration under rocprofv3 (kernel trace): python tools/prof_config.py C3 [--steps 3]

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/prof_config.py C3
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from pinnrl_amd import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--forward", action="store_true")
ap.add_argument("--points", type=int, default=0, help="override the configuration's point count (fixed-cost fits)")
args = ap.parse_args()
name, net, eq, n_req = B.CONFIGS[args.tag]()
if args.points:
    n_req = args.points
dev = B.dev
torch.manual_seed(1)
if args.tag == "C3":
    x = torch.rand(n_req, 1, device=dev) * 2 - 1
    t = torch.rand(n_req, 1, device=dev)
else:
    x, t = eq.generate_collocation_points(n_req, strategy="uniform")
N = x.shape[0]
prog, pd = net.program(), eq._pde_desc()
flat = E.new_flat_grad(prog, dev)
for _ in range(args.steps + 1):
    flat.zero_()
    if args.forward:
        E.residual_forward(prog, pd, x, t, want_residual=False)
    else:
        E.residual_loss_grad(prog, pd, x, t, 1.0 / N, flat)
torch.cuda.synchronize()
print(args.tag, name, N, "points")
