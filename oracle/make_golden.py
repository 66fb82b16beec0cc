"""Pin the oracle against the imported reference and write `tests/golden/*.npz`.

Run ONLY in the build container (the reference lives at /root/reference and
never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

For every case it (1) builds the reference `PINNModel` + `XxxEquation` under a
fixed seed, (2) asserts that `oracle.reference_path` reproduces theta_0, the
forward, `compute_derivatives`, the residual, mean(r^2) and dL/dtheta of the
reference (bit-for-bit where the op sequence is identical), and (3) stores the
inputs and the REFERENCE's outputs (fp32 and an fp64 twin) as a fixture.  The
fixtures are data only — no reference source text.
"""

from __future__ import annotations

import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from pinnrl.config import Config, ModelConfig  # noqa: E402  (reference)
from pinnrl.neural_networks import PINNModel  # noqa: E402
from pinnrl.pdes.allen_cahn import AllenCahnEquation  # noqa: E402
from pinnrl.pdes.black_scholes import BlackScholesEquation  # noqa: E402
from pinnrl.pdes.burgers_equation import BurgersEquation  # noqa: E402
from pinnrl.pdes.cahn_hilliard import CahnHilliardEquation  # noqa: E402
from pinnrl.pdes.convection_equation import ConvectionEquation  # noqa: E402
from pinnrl.pdes.heat_equation import HeatEquation  # noqa: E402
from pinnrl.pdes.kdv_equation import KdVEquation  # noqa: E402
from pinnrl.pdes.pde_base import PDEConfig  # noqa: E402
from pinnrl.pdes.pendulum_equation import PendulumEquation  # noqa: E402
from pinnrl.pdes.wave_equation import WaveEquation  # noqa: E402

from oracle import reference_path as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CPU = torch.device("cpu")

PDE_CLS = {
    "heat": HeatEquation,
    "burgers": BurgersEquation,
    "allen_cahn": AllenCahnEquation,
    "kdv": KdVEquation,
    "cahn_hilliard": CahnHilliardEquation,
    "wave": WaveEquation,
    "convection": ConvectionEquation,
    "black_scholes": BlackScholesEquation,
    "pendulum": PendulumEquation,
}

PDE_DEFAULTS = {
    # name: (domain, time_domain, parameters, initial_condition)
    "heat": ([(0.0, 1.0)], (0.0, 1.0), {"alpha": 0.01}, {"type": "sine", "amplitude": 1.0, "frequency": 2.0}),
    "burgers": ([(-1.0, 1.0)], (0.0, 1.0), {"nu": 0.01 / math.pi}, {"type": "sine", "amplitude": -1.0, "frequency": 1.0}),
    "allen_cahn": ([(-1.0, 1.0)], (0.0, 1.0), {"epsilon": 0.01}, {"type": "tanh", "epsilon": 0.1}),
    "kdv": ([(-15.0, 15.0)], (0.0, 5.0), {"speed": 1.0}, {"type": "soliton", "speed": 1.0}),
    "cahn_hilliard": ([(0.0, 1.0)], (0.0, 1.0), {"epsilon": 0.01}, {"type": "tanh"}),
    "wave": ([(0.0, 1.0)], (0.0, 1.0), {"c": 1.0}, {"type": "sine", "amplitude": 1.0, "frequency": 1.0}),
    "convection": ([(0.0, 2.0)], (0.0, 1.0), {"velocity": [1.0]}, {"type": "sine", "amplitude": 1.0, "frequency": 1.0}),
    "black_scholes": ([(0.0, 200.0)], (0.0, 1.0), {"sigma": 0.2, "r": 0.05}, {"type": "call_option", "strike_price": 100.0}),
    "pendulum": ([(0.0, 1.0)], (0.0, 10.0), {"g": 9.81, "L": 1.0}, {"type": "small_angle", "initial_angle": 0.5}),
}


def make_ref_model(spec: O.ArchSpec):
    cfg = Config.__new__(Config)
    cfg.device = CPU
    cfg.model = ModelConfig(
        input_dim=spec.input_dim,
        hidden_dim=spec.hidden_dim,
        output_dim=spec.output_dim,
        num_layers=spec.num_layers,
        activation=spec.activation,
        dropout=0.0,
        layer_norm=spec.layer_norm,
        architecture=spec.architecture,
    )
    cfg.model.mapping_size = spec.mapping_size
    cfg.model.scale = spec.scale
    cfg.model.omega_0 = spec.omega_0
    cfg.model.num_heads = spec.num_heads
    if spec.architecture == "resnet":
        cfg.model.num_blocks = spec.num_blocks if spec.num_blocks is not None else spec.num_layers
    cfg.model.device = CPU
    return PINNModel(cfg, device=CPU)


def make_ref_pde(pde: O.PdeSpec):
    cfg = PDEConfig(
        name=pde.name,
        domain=[tuple(d) for d in pde.domain],
        time_domain=tuple(pde.time_domain),
        parameters=dict(pde.parameters),
        boundary_conditions=dict(pde.boundary_conditions),
        initial_condition=dict(pde.initial_condition),
        exact_solution={},
        dimension=pde.dimension,
        device=CPU,
    )
    return PDE_CLS[pde.name](config=cfg)


def pde_spec(name: str, dimension: int = 1) -> O.PdeSpec:
    dom, td, par, ic = PDE_DEFAULTS[name]
    if dimension > 1:
        dom = [dom[0]] * dimension
    return O.PdeSpec(
        name=name,
        dimension=dimension,
        domain=dom,
        time_domain=td,
        parameters=par,
        boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}},
        initial_condition=ic,
    )


def points(pde: O.PdeSpec, ref_pde, n: int, seed: int):
    """`n` reference-sampled points + the domain corners (adversarial: clamp edges)."""
    torch.manual_seed(seed)
    x, t = ref_pde.generate_collocation_points(n, strategy="uniform")
    torch.manual_seed(seed)
    xo, to = O.sample_uniform(pde, n)
    assert torch.equal(x, xo) and torch.equal(t, to), "sample_uniform restatement differs from the reference"
    lo = torch.tensor([[d[0] for d in pde.domain[: pde.dimension]]], dtype=torch.float32)
    hi = torch.tensor([[d[1] for d in pde.domain[: pde.dimension]]], dtype=torch.float32)
    t0, t1 = pde.time_domain
    xc = torch.cat([lo, lo, hi, hi, (lo + hi) / 2], 0)
    tc = torch.tensor([[t0], [t1], [t0], [t1], [(t0 + t1) / 2]], dtype=torch.float32)
    return torch.cat([x, xc], 0).contiguous(), torch.cat([t, tc], 0).contiguous()


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(b.norm().item(), 1e-30))


def run_case(tag: str, spec: O.ArchSpec, pde: O.PdeSpec, n_pts: int, seed: int, manifest: dict):
    torch.manual_seed(seed)
    model = make_ref_model(spec)
    ref_pde = make_ref_pde(pde)
    sd_ref = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # (1) theta_0 restatement
    sd_o = O.init_state_dict(spec, seed=seed)
    assert list(sd_o.keys()) == list(sd_ref.keys()), (tag, list(sd_o.keys()), list(sd_ref.keys()))
    for k in sd_ref:
        assert torch.equal(sd_o[k], sd_ref[k]), f"{tag}: init of {k} differs"

    x, t = points(pde, ref_pde, n_pts, seed + 1)

    # (2) forward
    inp = torch.cat([x, t], 1)
    u_ref = model(inp).detach()
    u_o = O.network_forward(spec, sd_ref, inp)
    assert torch.equal(u_ref, u_o), f"{tag}: forward differs ({rel_l2(u_o, u_ref):.2e})"

    # (3) residual + loss + gradient, reference
    model.zero_grad()
    r_ref = ref_pde.compute_residual(model, x.clone(), t.clone())
    L_ref = ref_pde._apply_loss_fn(r_ref)
    L_ref.backward()
    names = [k for k, p in model.named_parameters()]
    g_ref = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}

    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, sd_ref, x.clone(), t.clone())
    e_r, e_L = rel_l2(r_o, r_ref.detach()), abs(float(L_o) - float(L_ref.detach())) / max(abs(float(L_ref.detach())), 1e-30)
    e_g = max(rel_l2(g_o[k], g_ref[k]) if g_ref[k].norm() > 0 else float(g_o[k].abs().max()) for k in names)
    assert e_r <= 1e-6 and e_L <= 1e-6 and e_g <= 2e-5, f"{tag}: restatement off: r {e_r:.2e} L {e_L:.2e} g {e_g:.2e}"

    # (4) derivative dictionary of the reference (jets) for the stream set this PDE uses
    jets = {}
    want_t, want_x = {"wave": [1, 2], "pendulum": [1, 2]}.get(pde.name, [1]), {
        "heat": [1, 2], "burgers": [1, 2], "allen_cahn": [1, 2], "kdv": [1, 2, 3], "cahn_hilliard": [1, 2, 3, 4],
        "wave": [1, 2], "convection": [1], "black_scholes": [1, 2], "pendulum": [],
    }[pde.name]
    if pde.dimension == 1:
        d = ref_pde.compute_derivatives(model, x, t, temporal_derivatives=want_t, spatial_derivatives=want_x)
        d_o = O.compute_derivatives(lambda z: O.network_forward(spec, sd_ref, z), x, t, want_t, want_x, 1)
        for k, v in d.items():
            assert torch.equal(v.detach(), d_o[k].detach()), f"{tag}: derivative {k} differs"
            jets["jet_" + k] = v.detach().numpy()
    else:
        d = ref_pde.compute_derivatives(model, x, t, temporal_derivatives=[1], spatial_derivatives=[])
        jets["jet_dt"] = d["dt"].detach().numpy()

    # (5) fp64 twin (the arbiter for tolerance-based parity)
    model64 = make_ref_model(spec).double()
    model64.load_state_dict({k: v.double() for k, v in sd_ref.items()})
    # Fourier keeps B as a buffer that forward() may re-place; make sure it is double
    for mod in model64.modules():
        if hasattr(mod, "B"):
            mod.B = mod.B.double()
    pde64 = make_ref_pde(pde)
    model64.zero_grad()
    r64 = pde64.compute_residual(model64, x.double(), t.double())
    L64 = pde64._apply_loss_fn(r64)
    L64.backward()
    g64 = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model64.named_parameters()}
    u64 = model64(inp.double()).detach()

    flat = lambda g: torch.cat([g[k].flatten() for k in names]).numpy()  # noqa: E731
    has_ln = spec.architecture in ("resnet", "attention") or (spec.architecture == "feedforward" and spec.layer_norm)
    exact = {}
    if has_ln:
        # The exact-derivative checker (oracle, composite LayerNorm, fp64).  Pinned here against the reference where
        # the reference is right: same residual (<= 1e-12), and the same gradient whenever the loss chains at most
        # two differentiations through a LayerNorm; with more, torch's fused layer_norm is wrong
        # and `grad64` (the reference's) is kept only as the witness of that error.
        sd64 = {k: v.double() for k, v in sd_ref.items()}
        r_c, L_c, g_c = O.residual_loss_and_grad(pde, spec, sd64, x.double(), t.double(), layer_norm="composite")
        e_rc = rel_l2(r_c, r64.detach())
        n_chain = {"kdv": 3, "cahn_hilliard": 4}.get(pde.name, 2) if pde.dimension == 1 else 1
        if n_chain <= 2:  # the residual itself chains <= 2 differentiations: the fused op is still right there
            assert e_rc <= 1e-11, f"{tag}: composite-LN residual differs from the reference's fp64 residual: {e_rc:.2e}"
        exact["residual64_exact"] = r_c.numpy()
        exact["loss64_exact"] = np.float64(L_c.item())
        exact["grad64_exact"] = torch.cat([g_c[k].flatten() for k in names]).numpy()
        exact["_witness"] = (rel_l2(torch.from_numpy(flat(g64)), torch.from_numpy(exact["grad64_exact"])), e_rc)
    arrays = {
        "x": x.numpy(), "t": t.numpy(), "u": u_ref.numpy(), "u64": u64.numpy(),
        "residual": r_ref.detach().numpy(), "residual64": r64.detach().numpy(),
        "loss": np.float32(L_ref.item()), "loss64": np.float64(L64.item()),
        "grad": flat(g_ref), "grad64": flat(g64),
    }
    arrays.update(jets)
    ln_err = exact.pop("_witness", None)
    arrays.update(exact)
    for k, v in sd_ref.items():
        arrays["sd::" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **arrays)
    manifest[tag] = {
        "seed": seed,
        "n_points": int(x.shape[0]),
        "arch": {k: getattr(spec, k) for k in spec.__dataclass_fields__},
        "pde": {"name": pde.name, "dimension": pde.dimension, "domain": [list(d) for d in pde.domain],
                "time_domain": list(pde.time_domain), "parameters": pde.parameters,
                "boundary_conditions": pde.boundary_conditions, "initial_condition": pde.initial_condition},
        "param_names": names,
        "fp32_vs_fp64": {"residual": rel_l2(r_ref.detach(), r64.detach()), "grad": rel_l2(torch.from_numpy(arrays["grad"]), torch.from_numpy(arrays["grad64"]))},
        "oracle_vs_reference": {"residual": e_r, "loss": e_L, "grad": e_g},
    }
    if ln_err is not None:  # torch's fused-LayerNorm third-derivative error, fp64 (witness; not a parity target)
        manifest[tag]["reference_grad_vs_exact"], manifest[tag]["reference_residual_vs_exact"] = ln_err
    print(f"{tag:38s} N={x.shape[0]:4d} params={sum(v.numel() for v in g_ref.values()):7d} "
          f"oracle-vs-ref r={e_r:.1e} g={e_g:.1e} | fp32-vs-fp64 r={manifest[tag]['fp32_vs_fp64']['residual']:.1e} "
          f"g={manifest[tag]['fp32_vs_fp64']['grad']:.1e}" + (f" | ref vs exact: grad {ln_err[0]:.1e} residual {ln_err[1]:.1e}" if ln_err is not None else ""))


def check_loss_terms():
    """`compute_loss` restatements (base class and HeatEquation's own) against the reference, value by value."""
    for name, fn, nb in (("burgers", O.compute_loss_terms, None), ("heat", O.compute_loss_terms_heat, 61)):
        pde = pde_spec(name)
        spec = O.ArchSpec("fourier", hidden_dim=32, num_layers=3, mapping_size=16, scale=4.0)
        torch.manual_seed(21)
        model = make_ref_model(spec)
        ref = make_ref_pde(pde)
        x, t = points(pde, ref, 123, 22)
        want = ref.compute_loss(model, x.clone(), t.clone())
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        got = fn(pde, lambda z: O.network_forward(spec, sd, z), x.clone(), t.clone())
        for k in ("residual", "boundary", "initial", "total"):
            a, b = float(got[k].detach()), float(want[k].detach())
            assert abs(a - b) <= 1e-6 * abs(b) + 1e-12, f"compute_loss[{name}][{k}]: oracle {a} vs reference {b}"
        print(f"compute_loss restatement == reference for {name}: " + ", ".join(f"{k}={float(want[k].detach()):.6g}" for k in ("residual", "boundary", "initial", "total")))


def check_data_modes(manifest: dict):
    """Data-driven modes (pde_base.py:281-291, 1187-1233; upstream tests/unit_tests/test_train_data_modes.py): the oracle's
    data term, mode gating and coefficient gradient against the reference, then a fixture of the reference's numbers."""
    from pinnrl.config import AdaptiveWeightsConfig, EarlyStoppingConfig, LearningRateSchedulerConfig, TrainingConfig

    def training(mode):
        return TrainingConfig(num_epochs=1, batch_size=8, num_collocation_points=8, num_boundary_points=4, num_initial_points=4,
                              learning_rate=1e-3, weight_decay=0.0, gradient_clipping=1.0,
                              early_stopping=EarlyStoppingConfig(enabled=False, patience=999, min_delta=1e-7),
                              learning_rate_scheduler=LearningRateSchedulerConfig(type="cosine", warmup_epochs=0, min_lr=1e-6, factor=0.5, patience=3),
                              adaptive_weights=AdaptiveWeightsConfig(enabled=False),
                              loss_weights={"residual": 1.0, "boundary": 10.0, "initial": 10.0, "data": 2.5}, mode=mode)

    spec = O.ArchSpec("fourier", hidden_dim=32, num_layers=3, mapping_size=16, scale=4.0)
    nu_true, nu_guess = 0.01 / math.pi, 0.05
    torch.manual_seed(41)
    model = make_ref_model(spec)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(42)
    x = torch.rand(97, 1, generator=gen) * 2 - 1
    t = torch.rand(97, 1, generator=gen)
    obs = {"x": torch.rand(53, 1, generator=gen) * 2 - 1, "t": torch.rand(53, 1, generator=gen)}
    obs["u"] = -torch.sin(math.pi * obs["x"]) * torch.exp(-obs["t"]) + 0.01 * torch.randn(53, 1, generator=gen)
    arrays = {"x": x.numpy(), "t": t.numpy(), "obs_x": obs["x"].numpy(), "obs_t": obs["t"].numpy(), "obs_u": obs["u"].numpy(),
              "nu_guess": np.float32(nu_guess)}
    for k, v in sd.items():
        arrays["sd/" + k] = v.numpy()
    names = [k for k, _ in model.named_parameters()]
    for mode in ("forward", "inverse", "data_only", "data_augmented"):
        inverse = mode == "inverse"
        cfg = PDEConfig(name="burgers", domain=[(-1.0, 1.0)], time_domain=(0.0, 1.0), parameters={"nu": nu_true},
                        boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}},
                        initial_condition={"type": "sine", "amplitude": -1.0, "frequency": 1.0}, exact_solution={}, dimension=1,
                        device=CPU, training=training(mode), trainable_parameters=["nu"] if inverse else [],
                        parameter_initial_guesses={"nu": nu_guess} if inverse else {},
                        observation_data={k: v.clone() for k, v in obs.items()})
        ref = BurgersEquation(config=cfg)
        model.zero_grad()
        want = ref.compute_loss(model, x.clone(), t.clone())
        plist = [p for _, p in model.named_parameters()] + (list(ref.trainable_parameters_iter()) if inverse else [])
        gw = torch.autograd.grad(want["total"], plist, allow_unused=True)
        # oracle
        params = {k: v.clone().requires_grad_(k != "model.fourier.B") for k, v in sd.items()}
        nu = torch.tensor(nu_guess if inverse else nu_true, requires_grad=inverse)
        pde = O.PdeSpec(name="burgers", parameters={"nu": nu if inverse else nu_true},
                        boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}},
                        initial_condition={"type": "sine", "amplitude": -1.0, "frequency": 1.0},
                        loss_weights={"residual": 1.0, "boundary": 10.0, "initial": 10.0, "data": 2.5})
        got = O.compute_loss_terms(pde, lambda z: O.network_forward(spec, params, z), x.clone(), t.clone(), observations=obs, mode=mode)
        go = torch.autograd.grad(got["total"], [params[k] for k in names] + ([nu] if inverse else []), allow_unused=True)
        for k in ("residual", "boundary", "initial", "data", "total"):
            a, b = float(got[k].detach()), float(want[k].detach())
            assert abs(a - b) <= 1e-6 * abs(b) + 1e-12, f"data modes[{mode}][{k}]: oracle {a} vs reference {b}"
            arrays[f"{mode}/{k}"] = np.float32(b)
        flat = lambda gs, ps: torch.cat([(g if g is not None else torch.zeros_like(p)).flatten() for g, p in zip(gs, ps)])  # noqa: E731
        gw_t, go_t = flat(gw[: len(names)], plist[: len(names)]), flat(go[: len(names)], [params[k] for k in names])
        assert rel_l2(go_t, gw_t) <= 1e-6, f"data modes[{mode}]: gradient {rel_l2(go_t, gw_t):.2e}"
        arrays[f"{mode}/grad"] = gw_t.numpy()
        if inverse:
            assert abs(float(go[-1]) - float(gw[-1])) <= 1e-6 * abs(float(gw[-1])), "d total / d nu"
            arrays["inverse/dnu"] = np.float32(float(gw[-1]))
        print(f"data mode {mode}: oracle == reference (total {float(want['total'].detach()):.6g}, data {float(want['data'].detach()):.6g})")
    np.savez(os.path.join(OUT, "data_modes.npz"), **arrays)
    manifest["_data_modes"] = {"modes": ["forward", "inverse", "data_only", "data_augmented"], "pde": "burgers", "arch": "fourier 3x32",
                              "points": 97, "observations": 53, "loss_weights": {"residual": 1.0, "boundary": 10.0, "initial": 10.0, "data": 2.5}}


LOSS_KINDS_IC = {
    "burgers": [{"type": "sine", "amplitude": -1.0, "frequency": 1.0}, {"type": "tanh", "epsilon": 0.1}],
    "heat": [{"type": "sin_exp_decay", "amplitude": 1.0, "frequency": 2.0}, {"type": "sine", "amplitude": 1.0, "frequency": 2.0}],
    "allen_cahn": [{"type": "tanh", "epsilon": 0.1}],
    "kdv": [{"type": "soliton", "speed": 1.0}],
    "cahn_hilliard": [{"type": "tanh"}],
    "wave": [{"type": "sine", "amplitude": 1.0, "frequency": 1.0}],
    "convection": [{"type": "sine", "amplitude": 1.0, "frequency": 1.0}],
    "black_scholes": [{"type": "call_option", "strike_price": 100.0}],
    "pendulum": [{"type": "small_angle", "initial_angle": 0.5}, {"type": "sine", "amplitude": 0.5, "frequency": 1.0},
                 {"type": "gaussian", "mean": 0.5, "std": 0.1}],
}
LOSS_KINDS_BC = [
    {"dirichlet": {"type": "fixed", "value": 0.0}},
    {"dirichlet": {"type": "fixed", "value": 0.5}},
    {"periodic": {}},
    {"neumann": {"type": "fixed", "value": 0.3}},
    {"left": {"type": "fixed", "value": 0.2}, "right": {"type": "fixed", "value": -0.1}},
]


def loss_kinds_fixture(manifest: dict):
    """`compute_loss` of all nine reference PDE classes under every deterministic initial-condition kind their own
    `_create_boundary_condition` accepts and five boundary-condition dictionaries: the REFERENCE's loss terms and
    d total / d theta for one small network, as a fixture the product's `compute_loss` is held to on the GPU
    (tests/test_loss_kinds_gpu.py).  A combination the reference refuses is recorded with the exception type: the product
    must refuse it too.  A combination whose terms depend on torch's RNG state is left out (CPU and device draws differ)."""
    spec = O.ArchSpec("fourier", hidden_dim=32, num_layers=3, mapping_size=16, scale=2.0)
    torch.manual_seed(51)
    model = make_ref_model(spec)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    arrays = {"sd/" + k: v.numpy() for k, v in sd.items()}
    combos = []
    for name, ics in LOSS_KINDS_IC.items():
        dom, td, par, _ = PDE_DEFAULTS[name]
        gen = torch.Generator().manual_seed(60 + len(combos))
        x = torch.rand(97, 1, generator=gen) * (dom[0][1] - dom[0][0]) + dom[0][0]
        t = torch.rand(97, 1, generator=gen) * (td[1] - td[0]) + td[0]
        arrays[f"{name}/x"], arrays[f"{name}/t"] = x.numpy(), t.numpy()
        for ic in ics:
            for bc in LOSS_KINDS_BC:
                key = f"{name}|{json.dumps(ic, sort_keys=True)}|{json.dumps(bc, sort_keys=True)}"
                entry = {"pde": name, "initial_condition": ic, "boundary_conditions": bc, "domain": [list(d) for d in dom],
                         "time_domain": list(td), "parameters": par, "key": key}
                try:
                    outs = []
                    for seed in (1, 2):  # RNG-dependent terms differ between the two
                        torch.manual_seed(seed)
                        ref = PDE_CLS[name](config=PDEConfig(name=name, domain=[tuple(d) for d in dom], time_domain=tuple(td),
                                                             parameters=dict(par), boundary_conditions=dict(bc), initial_condition=dict(ic),
                                                             exact_solution={}, dimension=1, device=CPU))
                        model.zero_grad()
                        want = ref.compute_loss(model, x.clone(), t.clone())
                        g = torch.autograd.grad(want["total"], [p for _, p in model.named_parameters()], allow_unused=True)
                        g = torch.cat([(gi if gi is not None else torch.zeros_like(p)).flatten() for gi, (_, p) in zip(g, model.named_parameters())])
                        outs.append(({k: float(v.detach()) for k, v in want.items() if torch.is_tensor(v) and v.numel() == 1}, g))
                except Exception as e:  # the reference refuses the combination
                    entry["raises"] = type(e).__name__
                    combos.append(entry)
                    continue
                (l1, g1), (l2, g2) = outs
                if any(abs(l1[k] - l2[k]) > 1e-7 * abs(l1[k]) for k in l1) or not torch.allclose(g1, g2, rtol=1e-6, atol=0):
                    entry["rng_dependent"] = True
                    combos.append(entry)
                    continue
                idx = len([c for c in combos if "index" in c])
                entry["index"] = idx
                for k, v in l1.items():
                    arrays[f"{idx}/{k}"] = np.float32(v)
                arrays[f"{idx}/grad"] = g1.numpy()
                combos.append(entry)
    np.savez_compressed(os.path.join(OUT, "loss_kinds.npz"), **arrays)
    manifest["_loss_kinds"] = {"param_names": names, "combos": combos,
                               "arch": {"architecture": "fourier", "hidden_dim": 32, "num_layers": 3, "mapping_size": 16, "scale": 2.0}}
    n_ok = len([c for c in combos if "index" in c])
    print(f"loss_kinds: {n_ok} combinations with numbers, {len([c for c in combos if 'raises' in c])} the reference refuses, "
          f"{len([c for c in combos if c.get('rng_dependent')])} RNG-dependent (left out)")


EXACT_KINDS = {
    "burgers": [{}, {"type": "cole_hopf"}, {"type": "cole_hopf", "viscosity": 0.05}, {"type": "tanh", "epsilon": 0.1}],
    "heat": [{}, {"type": "sin_exp_decay", "amplitude": 1.0, "frequency": 2.0}, {"type": "sine", "amplitude": 0.5, "frequency": 1.0}],
    "allen_cahn": [{}, {"type": "tanh"}],
    "kdv": [{}, {"type": "soliton", "speed": 1.0}],
    "cahn_hilliard": [{}, {"type": "tanh"}],
    "wave": [{}, {"type": "sine", "amplitude": 1.0, "frequency": 1.0}],
    "convection": [{}, {"type": "sine", "amplitude": 1.0, "frequency": 1.0}],
    "black_scholes": [{}, {"type": "call_option", "strike_price": 100.0}],
    "pendulum": [{}, {"type": "small_angle", "initial_angle": 0.5}, {"type": "sine", "amplitude": 0.5, "frequency": 1.0}],
}


def exact_solution_fixture(manifest: dict):
    """`exact_solution(x, t)` of the nine reference classes (what `validate` and the live-snapshot fields compare against) under
    every `exact_solution` dictionary kind each class branches on: the reference's values on 64 points, or the exception type it
    raises.  tests/test_exact_solutions_cpu.py holds the product's classes to them on a CPU device."""
    arrays, entries = {}, []
    for name, kinds in EXACT_KINDS.items():
        dom, td, par, ic = PDE_DEFAULTS[name]
        gen = torch.Generator().manual_seed(90 + len(entries))
        x = torch.rand(64, 1, generator=gen) * (dom[0][1] - dom[0][0]) + dom[0][0]
        t = torch.rand(64, 1, generator=gen) * (td[1] - td[0]) + td[0]
        arrays[f"{name}/x"], arrays[f"{name}/t"] = x.numpy(), t.numpy()
        for ex in kinds:
            e = {"pde": name, "exact_solution": ex, "domain": [list(d) for d in dom], "time_domain": list(td), "parameters": par,
                 "initial_condition": ic}
            try:
                ref = PDE_CLS[name](config=PDEConfig(name=name, domain=[tuple(d) for d in dom], time_domain=tuple(td), parameters=dict(par),
                                                     boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}},
                                                     initial_condition=dict(ic), exact_solution=dict(ex), dimension=1, device=CPU))
                try:
                    u = ref.exact_solution(x.clone(), t.clone())
                except RuntimeError:  # Burgers' Cole-Hopf form differentiates phi w.r.t. x by autograd: it needs x.requires_grad
                    u = ref.exact_solution(x.clone().requires_grad_(True), t.clone())
                    e["reference_needs_x_requires_grad"] = True
                if u is None:  # "if not self.config.exact_solution: return None"
                    e["returns_none"] = True
                else:
                    e["index"] = len([c for c in entries if "index" in c])
                    arrays[f'{e["index"]}/u'] = u.detach().numpy()
            except Exception as err:
                e["raises"] = type(err).__name__
            entries.append(e)
    # the pendulum class's closed-form helpers (pendulum_equation.py:232-289)
    pend = []
    dom, td, par, _ = PDE_DEFAULTS["pendulum"]
    x, t = torch.from_numpy(arrays["pendulum/x"]), torch.from_numpy(arrays["pendulum/t"])
    for ic in LOSS_KINDS_IC["pendulum"] + [{"type": "gaussian", "amplitude": 0.7, "center": 0.4, "sigma": 0.2}]:
        for bc in ({"dirichlet": {"type": "fixed", "value": 0.25}}, {"dirichlet": {"type": "periodic"}}):
            ref = PendulumEquation(config=PDEConfig(name="pendulum", domain=[tuple(d) for d in dom], time_domain=tuple(td), parameters=dict(par),
                                                    boundary_conditions=dict(bc), initial_condition=dict(ic), exact_solution={}, dimension=1, device=CPU))
            i = len(pend)
            arrays[f"pendulum_ic/{i}"] = ref.compute_initial_condition(x.clone()).numpy()
            arrays[f"pendulum_bc/{i}"] = ref.compute_boundary_condition(x.clone(), t.clone()).numpy()
            pend.append({"initial_condition": ic, "boundary_conditions": bc, "index": i})
    manifest["_pendulum_helpers"] = pend
    # HeatEquation.exact_solution_sine (heat_equation.py:197-212), 1-D and 2-D
    for dim in (1, 2):
        hd, htd, hpar, hic = PDE_DEFAULTS["heat"]
        ref = HeatEquation(config=PDEConfig(name="heat", domain=[tuple(hd[0])] * dim, time_domain=tuple(htd), parameters=dict(hpar),
                                            boundary_conditions={"dirichlet": {"type": "fixed", "value": 0.0}}, initial_condition=dict(hic),
                                            exact_solution={"type": "sine", "amplitude": 0.8, "frequency": 1.5}, dimension=dim, device=CPU))
        gen = torch.Generator().manual_seed(200 + dim)
        hx, ht = torch.rand(32, dim, generator=gen), torch.rand(32, 1, generator=gen)
        arrays[f"heat_sine/{dim}/x"], arrays[f"heat_sine/{dim}/t"] = hx.numpy(), ht.numpy()
        arrays[f"heat_sine/{dim}/u"] = ref.exact_solution_sine(hx.clone(), ht.clone()).numpy()
    np.savez_compressed(os.path.join(OUT, "exact_solutions.npz"), **arrays)
    manifest["_exact_solutions"] = entries
    print(f"exact_solutions: {len([c for c in entries if 'index' in c])} with values, {len([c for c in entries if c.get('returns_none')])} None, "
          f"{len([c for c in entries if 'raises' in c])} the reference raises on: {[(c['pde'], c['exact_solution'].get('type'), c['raises']) for c in entries if 'raises' in c]}")


def quirk_witnesses(manifest: dict):
    """The behavioural quirks of SURVEY §0.3/§0.4, pinned as data."""
    w = {}
    pde = pde_spec("heat")
    ref = make_ref_pde(pde)
    torch.manual_seed(7)
    x, t = ref.generate_collocation_points(5000, strategy="uniform")
    w["sample_uniform_5000_shape"] = list(x.shape)
    for n in (50000, 100000, 200000):
        w[f"sample_uniform_{n}_rows"] = int(np.sqrt(n)) ** 2
    torch.manual_seed(7)
    spec = O.ArchSpec(architecture="feedforward", hidden_dim=16, num_layers=2)
    model = make_ref_model(spec)
    d = ref.compute_derivatives(model, x[:64], t[:64], temporal_derivatives=[1], spatial_derivatives=[2])
    d12 = ref.compute_derivatives(model, x[:64], t[:64], temporal_derivatives=[1], spatial_derivatives=[1, 2])
    w["heat_dx2_equals_first_derivative"] = bool(torch.equal(d["dx2"], d12["dx"]))
    ch = pde_spec("cahn_hilliard", dimension=2)
    refch = make_ref_pde(ch)
    spec3 = O.ArchSpec(architecture="feedforward", hidden_dim=16, num_layers=2, input_dim=3)
    torch.manual_seed(7)
    m3 = make_ref_model(spec3)
    xx, tt = refch.generate_collocation_points(64, strategy="uniform")
    r = refch.compute_residual(m3, xx, tt)
    dd = refch.compute_derivatives(m3, xx, tt, temporal_derivatives=[1], spatial_derivatives=[])
    w["cahn_hilliard_2d_residual_is_u_t"] = bool(torch.equal(r.detach(), dd["dt"].detach()))
    manifest["_quirks"] = w
    print("quirks:", w)


def sampler_fixtures(manifest: dict):
    """Pin the sampler restatements (uniform 1-D / 2-D, stratified, RL-adaptive incl. the DQN policy network)
    against the imported reference under shared seeds, bit for bit, and store the reference's outputs."""
    from pinnrl.rl.rl_agent import RLAgent  # reference

    out, info = {}, {}
    burg = pde_spec("burgers")
    ch2 = pde_spec("cahn_hilliard", 2)
    for tag, pde, n in (("uniform1d", burg, 1000), ("uniform2d", ch2, 500), ("uniform2d_small", ch2, 20)):
        ref = make_ref_pde(pde)
        torch.manual_seed(31)
        x, t = ref.generate_collocation_points(n, strategy="uniform")
        torch.manual_seed(31)
        xo, to = O.sample_uniform(pde, n)
        assert torch.equal(x, xo) and torch.equal(t, to), tag
        out[tag + "_x"], out[tag + "_t"] = x.numpy(), t.numpy()
        info[tag] = {"pde": pde.name, "dimension": pde.dimension, "n": n, "seed": 31, "rows": int(x.shape[0])}
    for tag, pde, n in (("stratified1d", burg, 257), ("stratified2d", ch2, 64)):
        ref = make_ref_pde(pde)
        torch.manual_seed(32)
        x, t = ref.generate_collocation_points(n, strategy="stratified")
        torch.manual_seed(32)
        xo, to = O.sample_stratified(pde, n)
        assert torch.equal(x, xo) and torch.equal(t, to), tag
        out[tag + "_x"], out[tag + "_t"] = x.numpy(), t.numpy()
        info[tag] = {"pde": pde.name, "dimension": pde.dimension, "n": n, "seed": 32, "rows": int(x.shape[0])}
    # DQN policy: theta_0 and the scorer, then two consecutive adaptive draws (the second one decays epsilon)
    ac = pde_spec("allen_cahn")
    for tag, eps, n in (("adaptive_explore", 1.0, 400), ("adaptive_exploit", 0.0, 400), ("adaptive_exploit_big", 0.0, 20000)):
        torch.manual_seed(33)
        agent = RLAgent(state_dim=2, action_dim=1, hidden_dim=64, device=CPU)
        torch.manual_seed(33)
        ao = O.make_agent(2, 1, 64)
        sd_ref = agent.policy_net.state_dict()
        assert list(sd_ref.keys()) == list(ao.policy.keys())
        for k in sd_ref:
            assert torch.equal(sd_ref[k], ao.policy[k]), f"DQN init of {k} differs"
        agent.epsilon = eps
        ao.epsilon = eps
        ref = make_ref_pde(ac)
        ref.rl_agent = agent
        hist: list = []
        torch.manual_seed(34)
        draws_ref = [ref.generate_collocation_points(n, strategy="adaptive") for _ in range(2)]
        torch.manual_seed(34)
        draws_o = [O.sample_adaptive(ac, n, ao, hist) for _ in range(2)]
        for (x, t), (xo, to) in zip(draws_ref, draws_o):
            assert torch.equal(x, xo) and torch.equal(t, to), tag
        assert agent.epsilon == ao.epsilon
        for i, (x, t) in enumerate(draws_ref):
            out[f"{tag}_x{i}"], out[f"{tag}_t{i}"] = x.numpy(), t.numpy()
        info[tag] = {"pde": "allen_cahn", "n": n, "agent_seed": 33, "draw_seed": 34, "epsilon_start": eps,
                     "epsilon_after": agent.epsilon, "rows": int(draws_ref[0][0].shape[0])}
        if tag == "adaptive_explore":
            for k, v in sd_ref.items():
                out["dqn::" + k] = v.numpy()
            agent.policy_net.eval()
            pts = torch.rand(17, 2)
            out["dqn_in"], out["dqn_out_eval"] = pts.numpy(), agent.policy_net(pts).detach().numpy()
            assert torch.equal(agent.policy_net(pts), O.dqn_forward(ao.policy, pts, training=False))
            agent.policy_net.train()
    np.savez_compressed(os.path.join(OUT, "samplers.npz"), **out)
    manifest["_samplers"] = info
    print("samplers:", {k: v["rows"] for k, v in info.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest: dict = {}
    A = O.ArchSpec
    cases = [
        ("burgers_fourier_4x128", A("fourier", hidden_dim=128, num_layers=4), pde_spec("burgers"), 251, 0),
        ("heat_fourier_4x128", A("fourier", hidden_dim=128, num_layers=4), pde_spec("heat"), 251, 10),
        ("burgers_fourier_3x32", A("fourier", hidden_dim=32, num_layers=3, mapping_size=16, scale=4.0), pde_spec("burgers"), 123, 1),
        ("burgers_feedforward_3x32", A("feedforward", hidden_dim=32, num_layers=3), pde_spec("burgers"), 123, 2),
        ("burgers_feedforward_4x128", A("feedforward", hidden_dim=128, num_layers=4), pde_spec("burgers"), 251, 12),
        ("heat_feedforward_gelu_3x64", A("feedforward", hidden_dim=64, num_layers=3, activation="gelu"), pde_spec("heat"), 123, 3),
        ("allen_cahn_feedforward_sigmoid_3x32", A("feedforward", hidden_dim=32, num_layers=3, activation="sigmoid"), pde_spec("allen_cahn"), 123, 4),
        ("kdv_siren_3x32", A("siren", hidden_dim=32, num_layers=3, omega_0=30.0), pde_spec("kdv"), 123, 5),
        ("kdv_siren_4x128", A("siren", hidden_dim=128, num_layers=4, omega_0=30.0), pde_spec("kdv"), 251, 15),
        ("allen_cahn_resnet_2x32", A("resnet", hidden_dim=32, num_layers=2, num_blocks=2), pde_spec("allen_cahn"), 123, 6),
        ("allen_cahn_resnet_3x128", A("resnet", hidden_dim=128, num_layers=3, num_blocks=3), pde_spec("allen_cahn"), 251, 16),
        ("cahn_hilliard2d_attention_2x32", A("attention", input_dim=3, hidden_dim=32, num_layers=2, activation="gelu"), pde_spec("cahn_hilliard", 2), 120, 7),
        ("cahn_hilliard1d_feedforward_3x32", A("feedforward", hidden_dim=32, num_layers=3), pde_spec("cahn_hilliard"), 123, 8),
        ("wave_feedforward_3x32", A("feedforward", hidden_dim=32, num_layers=3), pde_spec("wave"), 123, 9),
        ("convection_fourier_3x32", A("fourier", hidden_dim=32, num_layers=3), pde_spec("convection"), 123, 11),
        ("black_scholes_feedforward_3x32", A("feedforward", hidden_dim=32, num_layers=3), pde_spec("black_scholes"), 123, 13),
        ("pendulum_siren_3x32", A("siren", hidden_dim=32, num_layers=3, omega_0=30.0), pde_spec("pendulum"), 123, 14),
        # round 2: LayerNorm in the plain feedforward stack (the YAML / direct-dict default), LayerNorm under 3rd / 4th
        # order and second-time-derivative residuals, attention under a second-order 1-D residual, width 124
        ("burgers_feedforward_ln_3x32", A("feedforward", hidden_dim=32, num_layers=3, layer_norm=True), pde_spec("burgers"), 123, 20),
        ("heat_feedforward_ln_gelu_3x64", A("feedforward", hidden_dim=64, num_layers=3, activation="gelu", layer_norm=True), pde_spec("heat"), 123, 21),
        ("kdv_resnet_2x32", A("resnet", hidden_dim=32, num_layers=2, num_blocks=2), pde_spec("kdv"), 123, 22),
        ("cahn_hilliard1d_resnet_2x32", A("resnet", hidden_dim=32, num_layers=2, num_blocks=2), pde_spec("cahn_hilliard"), 123, 23),
        ("kdv_attention_2x32", A("attention", hidden_dim=32, num_layers=2, activation="gelu"), pde_spec("kdv"), 123, 29),
        ("burgers_attention_2x32", A("attention", hidden_dim=32, num_layers=2, activation="gelu"), pde_spec("burgers"), 123, 24),
        ("wave_resnet_2x32", A("resnet", hidden_dim=32, num_layers=2, num_blocks=2), pde_spec("wave"), 123, 25),
        ("burgers_feedforward_3x124", A("feedforward", hidden_dim=124, num_layers=3), pde_spec("burgers"), 123, 26),
        ("allen_cahn_resnet_2x124", A("resnet", hidden_dim=124, num_layers=2, num_blocks=2), pde_spec("allen_cahn"), 123, 27),
        ("kdv_feedforward_ln_3x124", A("feedforward", hidden_dim=124, num_layers=3, layer_norm=True), pde_spec("kdv"), 123, 28),
    ]
    for tag, spec, pde, n, seed in cases:
        run_case(tag, spec, pde, n, seed, manifest)
    check_loss_terms()
    check_data_modes(manifest)
    quirk_witnesses(manifest)
    sampler_fixtures(manifest)
    loss_kinds_fixture(manifest)
    exact_solution_fixture(manifest)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True, default=float)
    print("wrote", len(cases), "fixtures to", OUT)


if __name__ == "__main__":
    main()
