"""The CPU oracle reproduces every golden vector captured from the reference (not-gpu)."""

import numpy as np
import pytest
import torch

from conftest import CASES, MANIFEST, load_case, rel_l2
import oracle as O


@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_reference_vectors(tag):
    spec, pde, sd, a, m = load_case(tag)
    x, t = torch.from_numpy(a["x"]), torch.from_numpy(a["t"])
    u = O.network_forward(spec, sd, torch.cat([x, t], 1))
    assert torch.equal(u, torch.from_numpy(a["u"]))
    r, L, g = O.residual_loss_and_grad(pde, spec, sd, x, t)
    # same op sequence on the same torch build -> bitwise; allow 1e-6 for other hosts' BLAS
    assert rel_l2(r, a["residual"]) <= 1e-6
    assert abs(float(L) - float(a["loss"])) <= 1e-6 * abs(float(a["loss"])) + 1e-30
    flat = torch.cat([g[k].flatten() for k in m["param_names"]])
    assert rel_l2(flat, a["grad"]) <= 2e-5
    # fp32 reference vs its own fp64 twin: the noise floor the 1e-5 parity bar sits above
    assert rel_l2(a["residual"], a["residual64"]) < 1e-5


@pytest.mark.parametrize("tag", CASES)
def test_oracle_theta0_matches_reference_init(tag):
    spec, pde, sd, a, m = load_case(tag)
    sd0 = O.init_state_dict(spec, seed=m["seed"])
    assert list(sd0) == list(sd)
    for k in sd:
        assert torch.equal(sd0[k], sd[k]), k


def test_quirk_witnesses():
    q = MANIFEST["_quirks"]
    assert q["sample_uniform_5000_shape"] == [4900, 1]
    assert q["heat_dx2_equals_first_derivative"] is True
    assert q["cahn_hilliard_2d_residual_is_u_t"] is True
    pde = O.PdeSpec(name="heat", domain=[(0.0, 1.0)])
    torch.manual_seed(0)
    x, t = O.sample_uniform(pde, 5000)
    assert x.shape == (4900, 1) and t.shape == (4900, 1)
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0


def test_derivative_order_limits():
    f = lambda z: z.sum(1, keepdim=True)  # noqa: E731
    x, t = torch.zeros(4, 1), torch.zeros(4, 1)
    with pytest.raises(ValueError):
        O.compute_derivatives(f, x, t, temporal_derivatives=[3])
    with pytest.raises(ValueError):
        O.compute_derivatives(f, x, t, spatial_derivatives=[5])


def test_apply_loss_fn_identities():
    e = torch.linspace(-2, 2, 41).reshape(-1, 1)
    assert torch.allclose(O.apply_loss_fn(e, "mse"), (e**2).mean())
    assert torch.allclose(O.apply_loss_fn(e, "mae"), e.abs().mean())
    assert torch.allclose(O.apply_loss_fn(e, "huber", 0.5), torch.nn.functional.huber_loss(e, torch.zeros_like(e), delta=0.5))


# ---------------------------------------------------------------------------------------------
# The exact-derivative checker for networks with a LayerNorm (oracle, layer_norm="composite")
# ---------------------------------------------------------------------------------------------
LN_CASES = [c for c in CASES if "grad64_exact" in np.load(f"{__import__('conftest').GOLDEN}/{c}.npz").files]


def _jets64(spec, sd64, x, t, mode, orders=(1, 2, 3)):
    x = x.clone().requires_grad_(True)
    t = t.clone().requires_grad_(True)
    u = O.network_forward(spec, sd64, torch.cat([x, t], 1), layer_norm=mode)
    out, cur = [u], u
    for _ in orders:
        cur = torch.autograd.grad(cur, x, torch.ones_like(cur), create_graph=True)[0]
        out.append(cur)
    ut = torch.autograd.grad(u, t, torch.ones_like(u), create_graph=True)[0]
    return [o.detach() for o in out], ut.detach()


@pytest.mark.parametrize("tag", [c for c in LN_CASES if MANIFEST[c]["pde"]["dimension"] == 1])
def test_composite_layer_norm_equals_fused_up_to_second_input_derivatives(tag):
    """Same function, two op sequences: u, u_t, u_x, u_xx agree to fp64 rounding; the THIRD differentiation through
    torch's fused layer_norm does not (that is the library error the composite path exists to avoid)."""
    spec, pde, sd, a, m = load_case(tag)
    sd64 = {k: v.double() for k, v in sd.items()}
    x, t = torch.from_numpy(a["x"]).double(), torch.from_numpy(a["t"]).double()
    (u_f, ux_f, uxx_f, uxxx_f), ut_f = _jets64(spec, sd64, x, t, "fused")
    (u_c, ux_c, uxx_c, uxxx_c), ut_c = _jets64(spec, sd64, x, t, "composite")
    for name, f, c in (("u", u_f, u_c), ("u_t", ut_f, ut_c), ("u_x", ux_f, ux_c), ("u_xx", uxx_f, uxx_c)):
        assert rel_l2(f, c) <= 1e-11, (name, rel_l2(f, c))
    assert rel_l2(uxxx_f, uxxx_c) > 1e-6, "the fused op's third derivative is expected to be off on this torch build"


@pytest.mark.parametrize("tag", ["allen_cahn_resnet_2x32", "burgers_feedforward_ln_3x32", "kdv_resnet_2x32"])
def test_composite_gradient_agrees_with_finite_differences(tag):
    """d(mean r^2)/d(theta) of the composite path == central differences of its own loss (fp64), entry by entry
    on a sample of parameters from every tensor; the fused path's gradient does not pass this check."""
    spec, pde, sd, a, m = load_case(tag)
    sd64 = {k: v.double() for k, v in sd.items()}
    x, t = torch.from_numpy(a["x"][:48]).double(), torch.from_numpy(a["t"][:48]).double()
    _, _, g = O.residual_loss_and_grad(pde, spec, sd64, x, t, layer_norm="composite")
    _, _, g_fused = O.residual_loss_and_grad(pde, spec, sd64, x, t, layer_norm="fused")

    def loss(params):
        r = O.compute_residual(pde, lambda z: O.network_forward(spec, params, z, "composite"), x, t)
        return float((r.detach() ** 2).mean())

    gen = torch.Generator().manual_seed(0)
    num, ana, ana_fused = [], [], []
    for k in m["param_names"]:
        flat_idx = int(torch.randint(0, sd64[k].numel(), (1,), generator=gen))
        h = 1e-6 * max(1.0, float(sd64[k].flatten()[flat_idx].abs()))
        plus, minus = dict(sd64), dict(sd64)
        plus[k] = sd64[k].clone()
        plus[k].view(-1)[flat_idx] += h
        minus[k] = sd64[k].clone()
        minus[k].view(-1)[flat_idx] -= h
        num.append((loss(plus) - loss(minus)) / (2 * h))
        ana.append(float(g[k].flatten()[flat_idx]))
        ana_fused.append(float(g_fused[k].flatten()[flat_idx]))
    num, ana, ana_fused = (torch.tensor(v, dtype=torch.float64) for v in (num, ana, ana_fused))
    assert rel_l2(ana, num) <= 1e-6, rel_l2(ana, num)
    assert rel_l2(ana_fused, num) > 1e-5  # witness of the library error


@pytest.mark.parametrize("tag", LN_CASES)
def test_exact_fixtures_are_the_composite_oracle(tag):
    """`residual64_exact` / `grad64_exact` stored by make_golden.py == the composite oracle recomputed here; where the
    residual chains at most two differentiations it is also the reference's own fp64 residual."""
    spec, pde, sd, a, m = load_case(tag)
    sd64 = {k: v.double() for k, v in sd.items()}
    x, t = torch.from_numpy(a["x"]).double(), torch.from_numpy(a["t"]).double()
    r, L, g = O.residual_loss_and_grad(pde, spec, sd64, x, t, layer_norm="composite")
    assert rel_l2(r, a["residual64_exact"]) <= 1e-12
    flat = torch.cat([g[k].flatten() for k in m["param_names"]])
    assert rel_l2(flat, a["grad64_exact"]) <= 1e-10
    if m["reference_residual_vs_exact"] <= 1e-10:
        assert rel_l2(a["residual64"], a["residual64_exact"]) <= 1e-10
    if m["reference_grad_vs_exact"] <= 1e-10:  # at most two chained differentiations: the reference is exact too
        assert rel_l2(a["grad64"], a["grad64_exact"]) <= 1e-10
