"""Full training step (SURVEY §8(d) "secondary: full-step pts/s", §8(f) rank 1) on ONE GPU, for context.

    python tools/bench_step.py [--configs C1,C2,C3,C4] [--steps 30] [--modes autograd,manual,graph]

One step = fresh collocation sample (the configuration's own sampler: uniform, or the DQN-adaptive sampler for C3) ->
residual + boundary + initial loss terms -> gradient -> clip_grad_norm_ -> Adam, i.e. the reference's inner loop
(pinnrl/training/trainer.py:546-698), at the BASELINE configurations' full sizes:

    autograd  `PDETrainer.train_step` with torch autograd around the fused launches (the reference's call sequence)
    manual    the autograd-free launch list (`_manual_launches`: what `PDETrainer.train()` takes by itself)
    graph     the same list captured once in a HIP graph (`make_graphed_step`) and replayed
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from pinnrl_amd.config import Config, TrainingConfig  # noqa: E402
from pinnrl_amd.rl import RLAgent  # noqa: E402
from pinnrl_amd.training import PDETrainer  # noqa: E402


def build(tag):
    name, net, eq, n_req = B.CONFIGS[tag]()
    agent = None
    if tag == "C3":  # BASELINE C3: DQN adaptive sampling
        agent = RLAgent(state_dim=2, action_dim=1, hidden_dim=64, device=B.dev)
        eq.rl_agent = agent
    cfg = Config.__new__(Config)
    cfg.device = B.dev
    cfg.training = TrainingConfig()
    return name, net, eq, agent, cfg, n_req


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C3,C4")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--modes", default="autograd,manual,graph")
    args = ap.parse_args()
    print("| config | points | sampler | mode | ms/step | points/s | last total loss |")
    print("|---|---|---|---|---|---|---|")
    for tag in [c for c in args.configs.split(",") if c]:
        for mode in args.modes.split(","):
            torch.manual_seed(0)
            name, net, eq, agent, cfg, n_req = build(tag)
            tr = PDETrainer(net, eq, None, cfg, device=B.dev, rl_agent=agent, fast_step=False)
            if mode != "autograd":
                why = tr._manual_step_unsupported()
                if why is not None:
                    print(f"| {tag} | - | - | {mode} | not covered: {why} | | |")
                    continue
                tr._build_flat_state()
            if mode == "graph":
                run, losses = tr.make_graphed_step(n_req)
            else:
                def run(tr=tr, n_req=n_req):
                    x, t = tr._sample(n_req)
                    return tr.train_step(x, t)
                losses = run()
            n = int(tr._sample(n_req)[0].shape[0])
            for _ in range(3):
                out = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = run()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            last = losses if mode == "graph" else out
            sampler = "adaptive (DQN)" if agent is not None else cfg.training.collocation_distribution
            print(f"| {tag} {name} | {n} | {sampler} | {mode} | {ms:.3f} | {n / ms * 1e3:.3e} | {float(last['total'].detach()):.4e} |", flush=True)
            del tr, net, eq
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
