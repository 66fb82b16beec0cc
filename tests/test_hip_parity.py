"""GPU parity: the HIP path (through the C ABI) vs the golden vectors captured from the reference.

Tolerance: <= 1e-5 relative L2 against the reference's fp32 outputs AND against its fp64 twin
(north_star's bar); fp32 end to end, no reduced precision anywhere.
"""

import re

import pytest
import torch

from conftest import rel_err, CASES, load_case, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _arch(tag):
    return load_case(tag)[0].architecture


def _engine_cases():
    """(fixture, engine): the plain-MLP family runs through BOTH engines (the fused tile-major kernel where it
    applies, and the layer-major engine forced by PINN_FLAG_LAYER_MAJOR); LayerNorm architectures and widths the fused
    kernel does not take (124) run the layer-major engine either way."""
    out = []
    for c in CASES:
        spec = load_case(c)[0]
        out.append((c, "default"))
        if spec.architecture in ("fourier", "feedforward", "siren") and not spec.layer_norm and spec.hidden_dim % 32 == 0:
            out.append((c, "lm"))
    return out


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _exact_jets(spec, sd, x, t, NT, NX):
    """fp64 jets [u, d/dt.., d/dx..] from the oracle with composite LayerNorm (exact at every order)."""
    import oracle as O

    sd64 = {k: v.double() for k, v in sd.items()}
    x = x.double().clone().requires_grad_(True)
    t = t.double().clone().requires_grad_(True)
    u = O.network_forward(spec, sd64, torch.cat([x, t], 1), layer_norm="composite")
    out, cur = [u], u
    for _ in range(NT):
        cur = torch.autograd.grad(cur, t, torch.ones_like(cur), create_graph=True)[0]
        out.append(cur)
    cur = u
    for _ in range(NX):
        cur = torch.autograd.grad(cur, x, torch.ones_like(cur), create_graph=True)[0][:, 0:1]
        out.append(cur)
    return [o.detach() for o in out]


@pytest.mark.parametrize("tag,engine", _engine_cases())
def test_jets_residual_loss_and_gradient(tag, engine, dev):
    """Every fixture, every architecture: jets, residual, loss and dL/dtheta through the C ABI.

    Targets: the reference's vectors (fp32 and its fp64 twin) wherever the reference is exact; for networks with a
    LayerNorm the oracle's composite-LayerNorm arrays (`*_exact`, pinned by oracle/make_golden.py and
    tests/test_oracle_golden.py) — torch's fused layer_norm is wrong from the third chained differentiation on, so
    there the reference's own numbers are kept only as a bounded witness."""
    import jet_model as J
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E

    spec, pde, sd, a, m = load_case(tag)
    prog, names = program_from_spec(spec, sd, dev)
    if engine == "lm":
        prog.set_layer_major(True)
    pd = pde_desc_from_spec(pde)
    x, t = torch.from_numpy(a["x"]).to(dev), torch.from_numpy(a["t"]).to(dev)
    N = x.shape[0]
    NT, NX = J.pde_streams(pde.name, pde.dimension)
    exact = "grad64_exact" in a
    # jets
    jets = E.jets_forward(prog, x, t, NT, NX).cpu()
    assert rel_l2(jets[0], a["u64"]) <= TOL, "u vs fp64"
    want = _exact_jets(spec, sd, torch.from_numpy(a["x"]), torch.from_numpy(a["t"]), NT, NX if pde.dimension == 1 else 0)
    # Individual derivative streams of a LayerNorm network are differences of nearly equal terms (the component of
    # c_k along c_0 is projected out): the same formulas evaluated in fp32 on the CPU (tests/jet_model.py) sit at 1.4e-5
    # on u_xx of the attention fixture, although the residual built from them meets 1e-5.  The parity bar proper —
    # residual, loss, gradient at TOL — is asserted below; isolated streams of LayerNorm networks get 5e-5.
    jet_tol = 5e-5 if exact else 2 * TOL
    for s, w in enumerate(want):
        e_s = rel_l2(jets[s], w, label=f"jet stream {s}", tol=2 * TOL if s == 0 else jet_tol)
        assert e_s <= (2 * TOL if s == 0 else jet_tol), f"jet stream {s}: {e_s:.2e}"
    if not exact:  # the reference's own derivative dictionary (key = order requested)
        ref = {"jet_dt": 1, "jet_dt2": 2, "jet_dx": NT + 1, "jet_dx2": NT + 2, "jet_dx3": NT + 3, "jet_dx4": NT + 4}
        for k, s_ in ref.items():
            if k in a and s_ < jets.shape[0] and not (k.startswith("jet_dt") and int(k[-1] if k[-1].isdigit() else 1) > NT):
                assert rel_l2(jets[s_], a[k]) <= 2 * TOL, k  # reference fp32 3rd/4th derivatives carry ~3e-6 noise themselves
    r_key, L_key, g_key = ("residual64_exact", "loss64_exact", "grad64_exact") if exact else ("residual64", "loss64", "grad64")
    # forward-only call
    r, ssum = E.residual_forward(prog, pd, x, t)
    assert rel_l2(r.cpu(), a[r_key]) <= TOL, f"{rel_l2(r.cpu(), a[r_key]):.2e}"
    assert abs(float(ssum) / N - float(a[L_key])) <= TOL * abs(float(a[L_key]))
    if not exact:
        assert rel_l2(r.cpu(), a["residual"]) <= TOL
    # forward + reverse sweep
    flat = E.new_flat_grad(prog, dev)
    r2, s2 = E.residual_loss_grad(prog, pd, x, t, 1.0 / N, flat, want_residual=True)
    assert rel_l2(r2.cpu(), a[r_key], label="residual", tol=TOL) <= TOL
    assert rel_err(float(s2) / N, float(a[L_key]), label="loss", tol=TOL) <= TOL
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    got = torch.cat([by_name[k].flatten().cpu() for k in m["param_names"]])
    e_g = rel_l2(got, a[g_key], label="gradient", tol=TOL)
    assert e_g <= TOL, f"grad vs {g_key}: {e_g:.3e}"
    if not exact:
        assert rel_l2(got, a["grad"]) <= TOL
    elif m["reference_grad_vs_exact"] < 1e-3:  # witness: distance to the reference = torch's LayerNorm error, not ours
        assert rel_l2(got, a["grad64"]) <= 2 * m["reference_grad_vs_exact"] + TOL
    if spec.architecture == "attention":
        for k in m["param_names"]:
            if ".query." in k or ".key." in k:
                assert float(by_name[k].abs().max()) == 0.0  # dead parameters, zero gradient as in the reference
        if pde.dimension > 1:
            assert rel_l2(r.cpu(), jets[1].unsqueeze(1)) == 0.0  # quirk witness: the 2-D residual IS u_t


def test_build_info_and_fallback_units(dev):
    """pinn_build_info() names every wide-kernel unit that had to be built in the default MFMA form; each such unit
    still has to meet the parity bar (run one of them: wave needs stream set (2, 2))."""
    from pinnrl_amd import _lib

    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    info = _lib.build_info()
    print("build info:", info or "(all units in their preferred form)")
    assert "error" not in info.lower()
    units = sorted(set(re.findall(r"jet_wide_(\d)_(\d)_(\d)", info))) or [("2", "2", "0")]
    pde_of = {(1, 1): "heat", (1, 2): "burgers", (1, 3): "kdv", (1, 4): "cahn_hilliard", (2, 2): "wave", (2, 0): "pendulum"}
    act_of = {v: k for k, v in _lib.ACT.items()}
    for nt, nx, act in units:
        # width 128 is where the two MFMA forms differ most in register pressure; fourier covers encoder + MLP units
        if act_of[int(act)] == "sin":  # the sine units are SIREN's
            spec = O.ArchSpec("siren", hidden_dim=128, num_layers=3, omega_0=4.0)
        else:
            spec = O.ArchSpec("fourier", hidden_dim=128, num_layers=3, mapping_size=32, scale=3.0, activation=act_of[int(act)])
        pde = O.PdeSpec(name=pde_of[(int(nt), int(nx))], parameters={"c": 1.3, "alpha": 0.05, "epsilon": 0.05})
        sd = O.init_state_dict(spec, seed=int(nt) * 100 + int(nx) * 10 + int(act))
        torch.manual_seed(7)
        x, t = O.sample_uniform(pde, 400)
        r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, {k: v.double() for k, v in sd.items()}, x.double(), t.double())
        prog, names = program_from_spec(spec, sd, dev)
        flat = E.new_flat_grad(prog, dev)
        r, s = E.residual_loss_grad(prog, pde_desc_from_spec(pde), x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
        tol = TOL if act_of[int(act)] != "relu" else 1e-4  # relu kinks: a point within fp32 of 0 flips a branch
        assert rel_l2(r.cpu(), r_o) <= tol, (nt, nx, act, rel_l2(r.cpu(), r_o))
        by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
        keys = [k for k in g_o if k in by_name]
        got = torch.cat([by_name[k].flatten().cpu() for k in keys])
        want = torch.cat([g_o[k].flatten() for k in keys])
        assert rel_l2(got, want) <= tol, (nt, nx, act, rel_l2(got, want))


@pytest.mark.parametrize("tag", ["burgers_fourier_3x32", "burgers_feedforward_3x32", "kdv_siren_3x32"])
def test_autograd_jet_function_backward(tag, dev):
    """General path: jets returned to PyTorch, arbitrary downstream graph, cotangents come back."""
    import jet_model as J
    from hip_helpers import program_from_spec
    from pinnrl_amd import engine as E

    spec, pde, sd, a, m = load_case(tag)
    prog, names = program_from_spec(spec, sd, dev)
    for p, tr in zip(prog.tensors, prog.trainable):
        p.requires_grad_(tr)
    x, t = torch.from_numpy(a["x"]).to(dev), torch.from_numpy(a["t"]).to(dev)
    NT, NX = J.pde_streams(pde.name, pde.dimension)
    jets = E.JetFunction.apply(prog, x, t, NT, NX, *prog.tensors)
    r, _ = J.pde_residual(pde.name, pde.parameters, [jets[s].unsqueeze(1) for s in range(jets.shape[0])], x[:, 0:1], NT, NX)
    loss = (r**2).mean()
    loss.backward()
    by_name = {n: p.grad for n, p in zip(names, prog.tensors) if p.grad is not None}
    got = torch.cat([by_name[k].flatten().cpu() for k in m["param_names"]])
    assert rel_l2(got, a["grad64"]) <= TOL


def test_partial_tile_and_multi_tile_consistency(dev):
    """N not a multiple of the 32-point tile, and many tiles per workgroup: same per-point results."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec, pde, sd, a, m = load_case("burgers_fourier_3x32")
    prog, names = program_from_spec(spec, sd, dev)
    pd = pde_desc_from_spec(pde)
    torch.manual_seed(3)
    pde_big = O.PdeSpec(name="burgers", parameters=pde.parameters)
    xb, tb = O.sample_uniform(pde_big, 40000)  # 39 601 points = 1 238 tiles -> several per workgroup
    xb, tb = xb[:39601 - 7], tb[:39601 - 7]     # ragged tail
    r_all, s_all = E.residual_forward(prog, pd, xb.to(dev), tb.to(dev))
    for lo, hi in [(0, 1), (5, 37), (1000, 1033), (39000, xb.shape[0])]:
        r_part, _ = E.residual_forward(prog, pd, xb[lo:hi].to(dev), tb[lo:hi].to(dev))
        assert torch.equal(r_part.cpu(), r_all[lo:hi].cpu())
    r_o, L_o, g_o = O.residual_loss_and_grad(pde_big, spec, sd, xb, tb)
    assert rel_l2(r_all.cpu(), r_o) <= TOL
    flat = E.new_flat_grad(prog, dev)
    E.residual_loss_grad(prog, pd, xb.to(dev), tb.to(dev), 1.0 / xb.shape[0], flat)
    grads = E.split_flat_grad(prog, flat)
    by_name = {n: g for n, g in zip(names, grads) if g is not None}
    got = torch.cat([by_name[k].flatten().cpu() for k in m["param_names"]])
    want = torch.cat([g_o[k].flatten() for k in m["param_names"]])
    assert rel_l2(got, want) <= 5 * TOL  # fp32 oracle itself sums 39k terms in a different order
    assert abs(float(s_all) / xb.shape[0] - float(L_o)) <= 1e-5 * float(L_o)


@pytest.mark.parametrize("kernel", ["default", "lm"])
def test_fourier_feature_count_not_a_multiple_of_32(dev, kernel, monkeypatch):
    """mapping_size = 12 -> 24 Fourier features: the first MFMA layer's input width ends inside a 32-wide k-tile
    (pinned against the oracle on the fly; poisons LDS first so that never-written rows cannot pass as zeros)."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec = O.ArchSpec("fourier", hidden_dim=64, num_layers=3, mapping_size=12, scale=2.0)
    pde = O.PdeSpec(name="burgers", parameters={"nu": 0.02})
    sd = O.init_state_dict(spec, seed=5)
    torch.manual_seed(6)
    x, t = O.sample_uniform(pde, 500)
    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, sd, x, t)
    prog, names = program_from_spec(spec, sd, dev)
    prog.set_layer_major(kernel == "lm")
    pd = pde_desc_from_spec(pde)
    torch.full((1 << 22,), float("nan"), device=dev).sum()  # a NaN-laden kernel's registers / LDS precede ours
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
    assert rel_l2(r.cpu(), r_o) <= TOL
    assert abs(float(s) / x.shape[0] - float(L_o)) <= TOL * abs(float(L_o))
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    keys = [k for k in g_o if k in by_name]
    got = torch.cat([by_name[k].flatten().cpu() for k in keys])
    want = torch.cat([g_o[k].flatten() for k in keys])
    assert torch.isfinite(got).all() and rel_l2(got, want) <= TOL


def test_empty_and_single_point_inputs(dev):
    """N = 0 launches nothing and returns empty / zero results; N = 1 is one ragged tile."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E

    spec, pde, sd, a, m = load_case("burgers_fourier_3x32")
    prog, names = program_from_spec(spec, sd, dev)
    pd = pde_desc_from_spec(pde)
    x, t = torch.from_numpy(a["x"]).to(dev), torch.from_numpy(a["t"]).to(dev)
    x0, t0 = x[:0], t[:0]
    assert E.jets_forward(prog, x0, t0, 1, 2).shape == (4, 0)
    r, s = E.residual_forward(prog, pd, x0, t0)
    assert r.shape == (0, 1) and float(s) == 0.0
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x0, t0, 1.0, flat, want_residual=True)
    assert r.shape == (0, 1) and float(s) == 0.0 and float(flat.abs().max()) == 0.0
    r1, s1 = E.residual_forward(prog, pd, x[:1], t[:1])
    assert abs(float(r1) - float(a["residual64"][0])) <= 1e-5 * (1 + abs(float(a["residual64"][0])))
    assert abs(float(s1) - float(r1) ** 2) <= 1e-5 * float(r1) ** 2 + 1e-12


@pytest.mark.parametrize("arch,num_layers", [("fourier", 2), ("feedforward", 1), ("feedforward", 2), ("siren", 1)])
@pytest.mark.parametrize("kernel", ["default", "lm"])
def test_shallow_networks(arch, num_layers, kernel, dev, monkeypatch):
    """One MFMA layer (fourier with num_layers = 2, feedforward / siren with 2 hidden layers) or none at all
    (a single hidden layer: first Linear -> output Linear): the kernels' special-cased ends, against the fp64 oracle."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec = O.ArchSpec(arch, hidden_dim=64, num_layers=num_layers, mapping_size=16, scale=2.0, omega_0=5.0)
    pde = O.PdeSpec(name="burgers", parameters={"nu": 0.02})
    sd = O.init_state_dict(spec, seed=21)
    torch.manual_seed(22)
    x, t = O.sample_uniform(pde, 200)
    x, t = x[:150], t[:150]
    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, {k: v.double() for k, v in sd.items()}, x.double(), t.double())
    prog, names = program_from_spec(spec, sd, dev)
    prog.set_layer_major(kernel == "lm")
    pd = pde_desc_from_spec(pde)
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
    assert rel_l2(r.cpu(), r_o) <= TOL
    assert abs(float(s) / x.shape[0] - float(L_o)) <= TOL * abs(float(L_o))
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    keys = [k for k in g_o if k in by_name]
    got = torch.cat([by_name[k].flatten().cpu() for k in keys])
    want = torch.cat([g_o[k].flatten() for k in keys])
    assert rel_l2(got, want) <= TOL, f"{rel_l2(got, want):.3e}"


@pytest.mark.parametrize("loss,delta", [("mse", 1.0), ("mae", 1.0), ("huber", 0.3)])
def test_loss_functions_in_the_fused_epilogue(loss, delta, dev):
    """PDEBase._apply_loss_fn (pde_base.py:309-326): mean r^2 / mean |r| / Huber(delta), value and gradient."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec, pde, sd, a, m = load_case("burgers_fourier_3x32")
    pde = O.PdeSpec(name="burgers", parameters=pde.parameters, loss_function=loss, huber_delta=delta)
    x, t = torch.from_numpy(a["x"]), torch.from_numpy(a["t"])
    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, {k: v.double() for k, v in sd.items()}, x.double(), t.double())
    prog, names = program_from_spec(spec, sd, dev)
    pd = pde_desc_from_spec(pde)
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
    assert abs(float(s) / x.shape[0] - float(L_o)) <= TOL * abs(float(L_o))
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    keys = [k for k in g_o if k in by_name]
    got = torch.cat([by_name[k].flatten().cpu() for k in keys])
    want = torch.cat([g_o[k].flatten() for k in keys])
    assert rel_l2(got, want) <= 2 * TOL, f"{rel_l2(got, want):.3e}"


@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
def test_piecewise_linear_activations(act, dev):
    """relu / leaky_relu (base_network.py:91-104): second derivatives vanish identically, so Burgers reduces to
    u_t + u u_x.  Kinks: a pre-activation within rounding of zero may take the other branch than the fp64 oracle,
    hence the looser bound on this case only."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec = O.ArchSpec("feedforward", hidden_dim=64, num_layers=3, activation=act)
    pde = O.PdeSpec(name="burgers", parameters={"nu": 0.02})
    sd = O.init_state_dict(spec, seed=31)
    torch.manual_seed(32)
    x, t = O.sample_uniform(pde, 200)
    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, {k: v.double() for k, v in sd.items()}, x.double(), t.double())
    prog, names = program_from_spec(spec, sd, dev)
    pd = pde_desc_from_spec(pde)
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
    assert rel_l2(r.cpu(), r_o) <= 1e-4
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    keys = [k for k in g_o if k in by_name]
    got = torch.cat([by_name[k].flatten().cpu() for k in keys])
    want = torch.cat([g_o[k].flatten() for k in keys])
    assert rel_l2(got, want) <= 1e-4, f"{rel_l2(got, want):.3e}"


@pytest.mark.parametrize("arch,dim,pde_name", [("feedforward", 2, "heat"), ("fourier", 2, "cahn_hilliard"), ("siren", 3, "heat"),
                                               ("feedforward", 2, "black_scholes"), ("resnet", 2, "black_scholes")])
def test_multi_dimensional_inputs(arch, dim, pde_name, dev):
    """input_dim 3 and 4 (x, y[, z], t) through the MLP kernels; as in the reference, every spatial term of a >= 2-D
    residual vanishes, which the oracle reproduces."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import engine as E
    import oracle as O

    spec = O.ArchSpec(arch, input_dim=dim + 1, hidden_dim=64, num_layers=3, mapping_size=16, scale=2.0, omega_0=5.0)
    pde = O.PdeSpec(name=pde_name, dimension=dim, domain=((0.0, 1.0),) * dim, parameters={"alpha": 0.05, "epsilon": 0.05, "r": 0.07, "sigma": 0.3})
    sd = O.init_state_dict(spec, seed=51)
    torch.manual_seed(52)
    x, t = torch.rand(137, dim), torch.rand(137, 1)
    r_o, L_o, g_o = O.residual_loss_and_grad(pde, spec, {k: v.double() for k, v in sd.items()}, x.double(), t.double())
    prog, names = program_from_spec(spec, sd, dev)
    pd = pde_desc_from_spec(pde)
    flat = E.new_flat_grad(prog, dev)
    r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / x.shape[0], flat, want_residual=True)
    assert rel_l2(r.cpu(), r_o) <= TOL
    by_name = {n: g for n, g in zip(names, E.split_flat_grad(prog, flat)) if g is not None}
    keys = [k for k in g_o if k in by_name]
    got = torch.cat([by_name[k].flatten().cpu() for k in keys])
    want = torch.cat([g_o[k].flatten() for k in keys])
    assert rel_l2(got, want) <= TOL, f"{rel_l2(got, want):.3e}"


def test_cpu_tensors_are_refused():
    from pinnrl_amd import engine as E
    from hip_helpers import program_from_spec

    spec, pde, sd, a, m = load_case("burgers_fourier_3x32")
    prog, _ = program_from_spec(spec, sd, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.jets_forward(prog, torch.zeros(4, 1), torch.zeros(4, 1), 1, 2)


def test_misaligned_weight_view_is_refused_with_a_clear_error(dev):
    """a hidden weight that is not 16-byte aligned used to be re-routed silently to the layer-major engine, whose
    workspace the sizing call had not reported ("workspace too small").  Now: PINN_ERR_MISALIGNED naming the tensor."""
    from hip_helpers import pde_desc_from_spec, program_from_spec
    from pinnrl_amd import _lib
    from pinnrl_amd import engine as E

    spec, pde, sd, a, m = load_case("burgers_fourier_4x128")
    prog, names = program_from_spec(spec, sd, dev)
    k = names.index("model.layers.1.weight") if "model.layers.1.weight" in names else 3
    w = prog.tensors[k]
    buf = torch.zeros(w.numel() + 1, dtype=torch.float32, device=dev)
    buf[1:].copy_(w.flatten())
    prog.tensors[k] = buf[1:].view_as(w)  # same values, 4 bytes off a 16-byte boundary
    assert prog.tensors[k].data_ptr() % 16 == 4
    x, t = torch.from_numpy(a["x"]).to(dev), torch.from_numpy(a["t"]).to(dev)
    with pytest.raises(_lib.JetLibraryError, match="16-byte aligned"):
        E.residual_forward(prog, pde_desc_from_spec(pde), x, t)
    prog.set_layer_major(True)  # the packing engine takes any alignment
    r, _ = E.residual_forward(prog, pde_desc_from_spec(pde), x, t)
    assert rel_l2(r.cpu(), a["residual64"]) <= TOL
