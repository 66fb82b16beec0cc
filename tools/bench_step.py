"""Full training step (SURVEY §8(d) "secondary: full-step pts/s", §8(f) rank 1) on ONE GPU, for context.

    python tools/bench_step.py [--points 50000] [--steps 50] [--graph]

One step = fresh collocation sample -> compute_loss (fused residual launch + boundary + initial terms) -> backward ->
clip_grad_norm_ -> Adam, i.e. `PDETrainer.train_step` as the reference's loop runs it (trainer.py:546-698).
Every launch of the step except the fused residual is a small launch-bound kernel; with --graph the whole step is
captured once in a HIP graph (`PDETrainer.make_graphed_step`) and replayed.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import _burgers  # noqa: E402
from pinnrl_amd.config import TrainingConfig  # noqa: E402
from pinnrl_amd.training import PDETrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, model, pde = _burgers(dev, hidden=128, layers=4, mapping=32, scale=10.0)
    cfg.training = TrainingConfig()
    trainer = PDETrainer(model, pde, optimizer_config=None, config=cfg, device=dev)

    def step():
        x, t = pde.generate_collocation_points(args.points, strategy="uniform")
        return trainer.train_step(x, t)

    n = int(pde.generate_collocation_points(args.points)[0].shape[0])
    if args.graph:  # the captured step contains no autograd (safe after eager steps too)
        run, losses = trainer.make_graphed_step(args.points)
    else:
        run = step
        for _ in range(5):
            losses = step()
    torch.cuda.synchronize()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    tot = float(losses["total"].detach())
    print(f"full step ({'graph replay' if args.graph else 'eager'}): {n} points, {ms:.3f} ms/step, {n / ms * 1e3:.3e} points/s, "
          f"last total loss {tot:.4e}")


if __name__ == "__main__":
    main()
