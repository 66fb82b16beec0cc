"""Randomised consistency sweep of the two step forms of PDETrainer on the GPU (test infrastructure, not shipped):

    python tools/fuzz_trainer.py --seconds 300 --seed 0

For a random (PDE, architecture, width, depth, activation, batch size, loss weights, clipping) it builds two identical trainers
from the same theta_0, pins the same batches, runs three Adam steps through the autograd step (`compute_loss` -> backward ->
clip -> torch Adam: the reference's call sequence) and through the autograd-free launch list that `train()` takes by itself,
and compares the loss dictionaries of every step and theta at the end."""
import argparse
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from pinnrl_amd import pdes as P  # noqa: E402
from pinnrl_amd.config import Config, TrainingConfig  # noqa: E402
from pinnrl_amd.training import PDETrainer  # noqa: E402

PDES = {
    "burgers": (P.BurgersEquation, [(-1.0, 1.0)], (0.0, 1.0), {"nu": 0.02}, {"type": "sine"}),
    "heat": (P.HeatEquation, [(0.0, 1.0)], (0.0, 1.0), {"alpha": 0.05}, {"type": "sine"}),
    "allen_cahn": (P.AllenCahnEquation, [(-1.0, 1.0)], (0.0, 1.0), {"epsilon": 0.05}, {"type": "tanh", "epsilon": 0.1}),
    "kdv": (P.KdVEquation, [(-5.0, 5.0)], (0.0, 1.0), {"speed": 1.0}, {"type": "soliton"}),
    "cahn_hilliard": (P.CahnHilliardEquation, [(-1.0, 1.0)], (0.0, 1.0), {"epsilon": 0.05}, {"type": "tanh"}),
    "wave": (P.WaveEquation, [(-1.0, 1.0)], (0.0, 1.0), {"c": 1.0}, {"type": "sine"}),
    "convection": (P.ConvectionEquation, [(-1.0, 1.0)], (0.0, 1.0), {"velocity": [1.0]}, {"type": "sine"}),
    "black_scholes": (P.BlackScholesEquation, [(0.1, 2.0)], (0.0, 1.0), {"sigma": 0.2, "r": 0.05}, {"type": "call_option", "strike_price": 1.0}),
    "pendulum": (P.PendulumEquation, [(-1.0, 1.0)], (0.0, 1.0), {"g": 9.81, "L": 1.0}, {"type": "small_angle", "initial_angle": 0.5}),
}


def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-300))


def make(rng_state, name, arch, w, layers, act, extra, clip, lr):
    cls, dom, tdom, par, ic = PDES[name]
    net = B.model(arch, w, layers, act, **extra)  # seeds torch with 0: identical theta_0
    eq = B.pde(cls, dom, tdom, dict(par), dict(ic))
    cfg = Config.__new__(Config)
    cfg.device = B.dev
    cfg.training = TrainingConfig(learning_rate=lr, gradient_clipping=clip)
    return net, eq, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", action="store_true", help="second run = the step captured in a HIP graph (make_graphed_step) and replayed, "
                    "first run = the eager launch list: same pinned batch, warm-up step included")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    t_end = time.time() + args.seconds
    n_case, bad, skipped = 0, [], 0
    while time.time() < t_end:
        n_case += 1
        name = rng.choice(list(PDES))
        arch = rng.choice(["feedforward", "fourier", "siren", "resnet", "attention"])
        w = rng.choice([16, 32, 64, 96, 128, 160, 256])
        layers = rng.randint(1, 2) if arch in ("resnet", "attention") else rng.randint(2, 4)
        act = "tanh" if arch == "siren" else rng.choice(["tanh", "gelu", "sigmoid", "relu"])
        extra = {}
        if arch == "resnet":
            extra["num_blocks"] = rng.randint(1, 2)
        if arch == "attention":
            extra["num_heads"] = rng.choice([1, 2, 4])
        if arch == "siren":
            extra["omega_0"] = rng.choice([3.0, 6.0])
        if arch == "fourier":
            extra["mapping_size"] = rng.choice([16, 32, 64])
            extra["scale"] = 2.0
        n = rng.choice([64, 100, 257, 961, 2500])
        clip = rng.choice([0.0, 0.5, 1.0])
        lr = rng.choice([1e-3, 3e-3])
        tag = f"#{n_case} {name} {arch} w={w} L={layers} {act} {extra} N={n} clip={clip} lr={lr}"
        if args.graph:
            try:
                thetas = []
                for graphed in (False, True):
                    net, eq, cfg = make(None, name, arch, w, layers, act, extra, clip, lr)
                    tr = PDETrainer(net, eq, {}, cfg, device=B.dev)
                    why = tr._manual_step_unsupported()
                    if why is not None:
                        raise RuntimeError("launch list refuses: " + str(why))
                    tr._build_flat_state()
                    torch.manual_seed(1000 + n_case)
                    xb, tb = eq.generate_collocation_points(n, strategy="uniform")
                    tr._sample = lambda m, xb=xb, tb=tb: (xb, tb)
                    if graphed:
                        replay, losses = tr.make_graphed_step(int(xb.shape[0]), warmup=1)
                        for _ in range(2):
                            replay()
                        torch.cuda.synchronize()
                        if not math.isfinite(float(losses["total"])):
                            raise FloatingPointError("non-finite total loss after replay")
                    else:
                        for _ in range(3):  # the graphed run's warm-up step + two replays
                            tr.train_step(xb, tb)
                    thetas.append(torch.cat([p.detach().flatten().cpu() for p in net.parameters()]))
            except Exception as e:
                bad.append(f"{tag} raised {type(e).__name__}: {str(e)[:160]}")
                print(f"{tag}: FAIL raised {type(e).__name__}: {str(e)[:160]}", flush=True)
                continue
            e_t = rel(thetas[1], thetas[0])
            ok = math.isfinite(e_t) and e_t <= 2e-4
            print(f"{tag}: graph replay vs eager launch list theta {e_t:.1e} {'ok' if ok else 'FAIL'}", flush=True)
            if not ok:
                bad.append(f"{tag} graph vs eager theta {e_t:.2e}")
            continue
        try:
            runs = []
            for manual in (False, True):
                net, eq, cfg = make(None, name, arch, w, layers, act, extra, clip, lr)
                tr = PDETrainer(net, eq, {}, cfg, device=B.dev, fast_step=None if manual else False)
                if manual:
                    why = tr._manual_step_unsupported()
                    if why is not None:
                        raise RuntimeError("launch list refuses: " + str(why))
                    tr._build_flat_state()
                torch.manual_seed(1000 + n_case)
                batches = [eq.generate_collocation_points(n, strategy="uniform") for _ in range(3)]
                losses = []
                for xb, tb in batches:
                    torch.manual_seed(77)  # boundary / initial points drawn inside compute_loss: same in both runs
                    out = tr.train_step(xb, tb)
                    losses.append({k: float(v) for k, v in out.items() if torch.is_tensor(v) or isinstance(v, float)})
                theta = torch.cat([p.detach().flatten().cpu() for p in net.parameters()])
                runs.append((losses, theta))
        except Exception as e:
            skipped += 1
            print(f"{tag}: skipped ({type(e).__name__}: {str(e)[:120]})", flush=True)
            continue
        (la, ta), (lm, tm) = runs
        worst_l = 0.0
        for s in range(3):  # every term against the step's TOTAL: a component of 1e-9 beside a total of 1 carries no information
            scale = max(abs(la[s].get("total", 0.0)), 1e-12)
            for k in la[s]:
                if k in lm[s] and math.isfinite(la[s][k]):
                    worst_l = max(worst_l, abs(la[s][k] - lm[s][k]) / scale)
        e_t = rel(tm, ta)
        # three Adam steps divide by sqrt(v): where a gradient entry is near zero the two summation orders differ by more than 1e-5
        ok = math.isfinite(e_t) and e_t <= 2e-4 and worst_l <= 1e-4
        print(f"{tag}: losses {worst_l:.1e} theta {e_t:.1e} {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            bad.append(f"{tag} losses {worst_l:.2e} theta {e_t:.2e}")
            for s in range(3):
                print("      step", s, {k: (round(la[s][k], 8), round(lm[s].get(k, float('nan')), 8)) for k in la[s]}, flush=True)
    print(f"cases {n_case}, skipped {skipped}, failures {len(bad)}")
    for b in bad:
        print("FAIL", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
