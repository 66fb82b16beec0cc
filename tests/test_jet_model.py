"""The closed-form jet algorithm (what the HIP kernels implement) equals the autograd oracle (not-gpu, fp64)."""

import pytest
import sympy as sp
import torch

from conftest import CASES, load_case, rel_l2
import jet_model as J
import oracle as O

MLP = [c for c in CASES if load_case(c)[0].architecture in ("fourier", "feedforward", "siren")]


def _to64(sd):
    return {k: v.double() for k, v in sd.items()}


@pytest.mark.parametrize("act,expr", [
    ("tanh", lambda z: sp.tanh(z)),
    ("sin", lambda z: sp.sin(3 * z)),
    ("gelu", lambda z: z * (1 + sp.erf(z / sp.sqrt(2))) / 2),
    ("sigmoid", lambda z: 1 / (1 + sp.exp(-z))),
])
def test_activation_derivative_tables(act, expr):
    z = sp.symbols("z")
    e = expr(z)
    pts = torch.tensor([-2.3, -0.7, 0.0, 0.31, 1.9], dtype=torch.float64)
    f = J.act_derivs(act, 3.0, pts, 5)
    for k in range(6):
        fn = sp.lambdify(z, sp.diff(e, z, k), "math")
        want = torch.tensor([fn(float(v)) for v in pts], dtype=torch.float64)
        assert torch.allclose(f[k], want, rtol=1e-11, atol=1e-12), (act, k)


@pytest.mark.parametrize("tag", MLP)
def test_jets_and_gradient_match_autograd(tag):
    spec, pde, sd, a, m = load_case(tag)
    sd = _to64(sd)
    x, t = torch.from_numpy(a["x"]).double(), torch.from_numpy(a["t"]).double()
    NT, NX = J.pde_streams(pde.name, pde.dimension)
    prog = J.mlp_program(spec, sd)
    inp = torch.cat([x, t], 1)
    jets, tape = J.mlp_jets_forward(prog, inp, NT, NX)
    assert rel_l2(jets[0], a["u64"]) < 1e-12
    r, dr = J.pde_residual(pde.name, pde.parameters, jets, x[:, 0:1], NT, NX, pde.dimension)
    assert rel_l2(r, a["residual64"]) < 1e-10
    N = x.shape[0]
    ubar = [2.0 * r / N * d for d in dr]
    g = J.mlp_jets_backward(prog, tape, ubar, NT, NX)
    flat = torch.cat([g[k[len("") :]].flatten() for k in m["param_names"]])
    assert rel_l2(flat, a["grad64"]) < 1e-9
    # and the fp32 reference sits within the parity bar of this fp64 truth
    assert rel_l2(a["residual"], r) < 1e-5


def test_layernorm_jets_match_autograd():
    torch.manual_seed(0)
    N, H = 7, 16
    x = torch.randn(N, 1, dtype=torch.float64, requires_grad=True)
    W = torch.randn(H, 1, dtype=torch.float64)
    gma, bta = torch.randn(H, dtype=torch.float64), torch.randn(H, dtype=torch.float64)

    def f(xx):
        z = torch.sin(xx @ W.T) + 0.3 * xx
        return torch.nn.functional.layer_norm(z, (H,), gma, bta, 1e-5)

    y = f(x)
    d1 = torch.stack([torch.autograd.grad(y[:, j].sum(), x, create_graph=True)[0][:, 0] for j in range(H)], 1)
    d2 = torch.stack([torch.autograd.grad(d1[:, j].sum(), x, retain_graph=True)[0][:, 0] for j in range(H)], 1)
    xd = x.detach()
    z0 = torch.sin(xd @ W.T) + 0.3 * xd
    z1 = torch.cos(xd @ W.T) * W.T + 0.3
    z2 = -torch.sin(xd @ W.T) * W.T**2
    out = J.ln_fwd([z0, z1, z2], gma, bta, 1e-5, 0, 2)
    assert torch.allclose(out[0], y.detach(), atol=1e-12)
    assert torch.allclose(out[1], d1.detach(), atol=1e-10)
    assert torch.allclose(out[2], d2.detach(), atol=1e-9)
