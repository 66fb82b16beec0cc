"""Randomised sweep of `PDEBase.compute_derivatives` request sets on the GPU against `oracle.compute_derivatives` (fp64):
keys, shapes, values, the chaining rule (the key names the order REQUESTED, the value what was actually chained), 1-D / 2-D / 3-D
inputs, and the ValueErrors.  python tools/fuzz_derivs.py --seconds 120 --seed 0"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import oracle as O  # noqa: E402
import bench_configs as B  # noqa: E402
from pinnrl_amd import pdes as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    t_end = time.time() + args.seconds
    n_case, bad = 0, []
    while time.time() < t_end:
        n_case += 1
        dim = rng.choice([1, 1, 2, 3])
        arch = rng.choice(["feedforward", "fourier", "siren", "resnet", "attention"])
        w = rng.choice([16, 32, 64, 128])
        layers = rng.randint(1, 2) if arch in ("resnet", "attention") else rng.randint(2, 3)
        act = "tanh" if arch == "siren" else rng.choice(["tanh", "gelu", "sigmoid"])
        extra = {"input_dim": dim + 1}
        if arch == "resnet":
            extra["num_blocks"] = 1
        if arch == "attention":
            extra["num_heads"] = 2
        if arch == "siren":
            extra["omega_0"] = 3.0
        if arch == "fourier":
            extra["mapping_size"], extra["scale"] = 16, 2.0
        td = rng.choice([None, [], [1], [2], [1, 2], [0, 1], [3], [2, 1]])
        sdv = rng.choice([None, [], [1], [2], [1, 2], [3], [1, 3], [2, 4], [1, 2, 3, 4], [4], [5], [2, 1], [0, 2]])
        n = rng.choice([1, 33, 100, 257])
        tag = f"#{n_case} dim={dim} {arch} w={w} L={layers} {act} t={td} x={sdv} N={n}"
        net = B.model(arch, w, layers, act, **extra)
        eq = P.BurgersEquation(P.PDEConfig(name="burgers", domain=[(-1.0, 1.0)] * dim, time_domain=(0.0, 1.0), parameters={"nu": 0.02},
                                           boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}}, initial_condition={"type": "sine"},
                                           exact_solution={}, dimension=dim, device=B.dev))
        spec = O.ArchSpec(architecture=arch, input_dim=dim + 1, hidden_dim=w, num_layers=layers, activation=act,
                          **{k: v for k, v in extra.items() if k != "input_dim"})
        sd64 = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
        torch.manual_seed(n_case)
        x, t = torch.rand(n, dim) * 2 - 1, torch.rand(n, 1)
        want = err_o = None
        try:
            want = O.compute_derivatives(lambda z: O.network_forward(spec, sd64, z, "composite"), x.double(), t.double(),
                                         temporal_derivatives=td, spatial_derivatives=sdv, dimension=dim)
        except Exception as e:
            err_o = e
        got = err_p = None
        try:
            got = eq.compute_derivatives(net, x.to(B.dev), t.to(B.dev), temporal_derivatives=td, spatial_derivatives=sdv)
        except Exception as e:
            err_p = e
        if err_o is not None or err_p is not None:
            same = err_o is not None and err_p is not None and type(err_o).__name__ == type(err_p).__name__
            print(f"{tag}: oracle {type(err_o).__name__ if err_o else 'ok'} / product {type(err_p).__name__ if err_p else 'ok'} {'ok' if same else 'FAIL'}", flush=True)
            if not same:
                bad.append(f"{tag}: oracle {err_o!r:.120} product {err_p!r:.120}")
            continue
        kw, kg = {k for k in want if not k.startswith('_')}, set(got)  # '_x' / '_t': the oracle's own handles on its leaves
        msg, ok = "", True
        if kw != kg:
            ok, msg = False, f"keys differ: oracle-only {sorted(kw - kg)} product-only {sorted(kg - kw)}"
        else:
            worst = 0.0
            for k in kw:
                a, b = got[k].detach().cpu().double(), want[k].detach()
                if a.shape != b.shape:
                    ok, msg = False, f"{k}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
                    break
                e = float((a - b).norm() / max(float(b.norm()), 1e-30)) if float(b.norm()) > 1e-12 else float(a.norm())
                worst = max(worst, e)
            if ok and worst > 2e-4:  # four chained differentiations of a random net at a single point: 5e-5 happens
                ok, msg = False, f"values: worst {worst:.1e}"
            elif ok:
                msg = f"{len(kw)} keys, worst {worst:.1e}"
        print(f"{tag}: {msg} {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            bad.append(f"{tag}: {msg}")
    print(f"cases {n_case}, failures {len(bad)}")
    for b in bad[:30]:
        print("FAIL", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
