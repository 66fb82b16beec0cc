"""Executable model of the HIP kernels' ALGORITHM (closed-form jets, no autograd).

Test infrastructure: a line-by-line Python/torch mirror of what
`pinns-rl-pde_amd/csrc/jet_device.h` computes — forward-mode propagation of
K = 1 + NT + NX derivative streams through the network (Faa di Bruno per
activation) and the hand-derived reverse sweep — so that the formulas can be
checked against the autograd oracle on CPU (fp64) before they run on a GPU, and
so that a failing GPU test can be bisected layer by layer.

Stream order: [value, d/dt .. d^NT/dt^NT, d/dx .. d^NX/dx^NX].
"""

from __future__ import annotations

import math
from typing import Dict, List, Mapping, Sequence, Tuple

import torch

Tensor = torch.Tensor
SQRT1_2 = 0.7071067811865476
INV_SQRT_2PI = 0.3989422804014327


# ----------------------------------------------------------------------------
# activation derivatives f0..f_ord at z  (mirrors act_derivs<> in jet_device.h)
# ----------------------------------------------------------------------------
def act_derivs(act: str, param: float, z: Tensor, ord_: int) -> List[Tensor]:
    f: List[Tensor] = []
    if act == "tanh":
        y = torch.tanh(z)
        f1 = 1 - y * y
        y2 = y * y
        f = [y, f1, -2 * y * f1, f1 * (6 * y2 - 2), 8 * y * f1 * (2 - 3 * y2), 8 * f1 * (2 - 15 * y2 + 15 * y2 * y2)]
    elif act == "sin":
        w = param
        s, c = torch.sin(w * z), torch.cos(w * z)
        f = [s, w * c, -(w**2) * s, -(w**3) * c, (w**4) * s, (w**5) * c]
    elif act == "gelu":
        phi = torch.exp(-0.5 * z * z) * INV_SQRT_2PI
        Phi = 0.5 * (1 + torch.erf(z * SQRT1_2))
        z2 = z * z
        f = [z * Phi, Phi + z * phi, phi * (2 - z2), phi * z * (z2 - 4), phi * (-z2 * z2 + 7 * z2 - 4),
             phi * z * (z2 * z2 - 11 * z2 + 18)]
    elif act == "sigmoid":
        s = torch.sigmoid(z)
        f1 = s * (1 - s)
        f2 = f1 * (1 - 2 * s)
        f3 = f1 * (1 - 6 * f1)
        f4 = f2 * (1 - 12 * f1)
        f5 = f3 * (1 - 12 * f1) - 12 * f2 * f2
        f = [s, f1, f2, f3, f4, f5]
    elif act == "relu":
        m = (z > 0).to(z.dtype)
        zero = torch.zeros_like(z)
        f = [z * m, m, zero, zero, zero, zero]
    elif act == "leaky_relu":
        m = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.01))
        zero = torch.zeros_like(z)
        f = [z * m, m, zero, zero, zero, zero]
    elif act == "identity":
        zero = torch.zeros_like(z)
        f = [z, torch.ones_like(z), zero, zero, zero, zero]
    else:
        raise ValueError(act)
    return f[: ord_ + 1]


def _dir_fwd(f: Sequence[Tensor], z: Sequence[Tensor]) -> List[Tensor]:
    """Faa di Bruno, one direction: z = [z1..zm] -> [y1..ym] (m <= 4)."""
    m = len(z)
    y = []
    if m >= 1:
        y.append(f[1] * z[0])
    if m >= 2:
        y.append(f[2] * z[0] ** 2 + f[1] * z[1])
    if m >= 3:
        y.append(f[3] * z[0] ** 3 + 3 * f[2] * z[0] * z[1] + f[1] * z[2])
    if m >= 4:
        y.append(f[4] * z[0] ** 4 + 6 * f[3] * z[0] ** 2 * z[1] + 3 * f[2] * z[1] ** 2 + 4 * f[2] * z[0] * z[2] + f[1] * z[3])
    return y


def _dir_bwd(f: Sequence[Tensor], z: Sequence[Tensor], ab: Sequence[Tensor]) -> Tuple[Tensor, List[Tensor]]:
    """Adjoint of _dir_fwd: returns (contribution to zbar_0, [zbar_1..zbar_m])."""
    m = len(z)
    z0b = torch.zeros_like(z[0]) if m else 0
    zb = [torch.zeros_like(z[0]) for _ in range(m)]
    if m >= 1:
        z0b = z0b + f[2] * z[0] * ab[0]
        zb[0] = zb[0] + f[1] * ab[0]
    if m >= 2:
        z0b = z0b + (f[3] * z[0] ** 2 + f[2] * z[1]) * ab[1]
        zb[0] = zb[0] + 2 * f[2] * z[0] * ab[1]
        zb[1] = zb[1] + f[1] * ab[1]
    if m >= 3:
        z0b = z0b + (f[4] * z[0] ** 3 + 3 * f[3] * z[0] * z[1] + f[2] * z[2]) * ab[2]
        zb[0] = zb[0] + (3 * f[3] * z[0] ** 2 + 3 * f[2] * z[1]) * ab[2]
        zb[1] = zb[1] + 3 * f[2] * z[0] * ab[2]
        zb[2] = zb[2] + f[1] * ab[2]
    if m >= 4:
        z0b = z0b + (f[5] * z[0] ** 4 + 6 * f[4] * z[0] ** 2 * z[1] + 3 * f[3] * z[1] ** 2 + 4 * f[3] * z[0] * z[2] + f[2] * z[3]) * ab[3]
        zb[0] = zb[0] + (4 * f[4] * z[0] ** 3 + 12 * f[3] * z[0] * z[1] + 4 * f[2] * z[2]) * ab[3]
        zb[1] = zb[1] + (6 * f[3] * z[0] ** 2 + 6 * f[2] * z[1]) * ab[3]
        zb[2] = zb[2] + 4 * f[2] * z[0] * ab[3]
        zb[3] = zb[3] + f[1] * ab[3]
    return z0b, zb


def act_fwd(act: str, param: float, z: List[Tensor], NT: int, NX: int) -> List[Tensor]:
    m = max(NT, NX)
    f = act_derivs(act, param, z[0], m)
    return [f[0]] + _dir_fwd(f, z[1 : 1 + NT]) + _dir_fwd(f, z[1 + NT : 1 + NT + NX])


def act_bwd(act: str, param: float, z: List[Tensor], ab: List[Tensor], NT: int, NX: int) -> List[Tensor]:
    m = max(NT, NX)
    f = act_derivs(act, param, z[0], m + 1)
    z0b = f[1] * ab[0]
    ct, zt = _dir_bwd(f, z[1 : 1 + NT], ab[1 : 1 + NT])
    cx, zx = _dir_bwd(f, z[1 + NT : 1 + NT + NX], ab[1 + NT : 1 + NT + NX])
    return [z0b + ct + cx] + zt + zx


# ----------------------------------------------------------------------------
# LayerNorm jets (per point, reduction over the feature axis)
# ----------------------------------------------------------------------------
def _ln_dir_fwd(c0, r, v, cs: Sequence[Tensor]):
    """One direction.  c0 centred value, r = v^-1/2, cs = centred derivative streams [c1..cm] (m <= 2 supported)."""
    m = len(cs)
    H = c0.shape[-1]
    out = []
    if m >= 1:
        v1 = 2 * (c0 * cs[0]).mean(-1, keepdim=True)
        r1 = -0.5 * r**3 * v1
        out.append(cs[0] * r + c0 * r1)
    if m >= 2:
        v2 = 2 * (cs[0] * cs[0] + c0 * cs[1]).mean(-1, keepdim=True)
        r2 = 0.75 * r**5 * v1 * v1 - 0.5 * r**3 * v2
        out.append(cs[1] * r + 2 * cs[0] * r1 + c0 * r2)
    if m >= 3:
        raise NotImplementedError("LayerNorm jets above 2nd order")
    return out


def ln_fwd(z: List[Tensor], gamma: Tensor, beta: Tensor, eps: float, NT: int, NX: int) -> List[Tensor]:
    """z streams: (N, H).  y_s = jets of ((z - mean) * rsqrt(var + eps)) * gamma + beta."""
    c = [s - s.mean(-1, keepdim=True) for s in z]
    v = (c[0] * c[0]).mean(-1, keepdim=True) + eps
    r = v**-0.5
    out = [c[0] * r * gamma + beta]
    for d in _ln_dir_fwd(c[0], r, v, c[1 : 1 + NT]):
        out.append(d * gamma)
    for d in _ln_dir_fwd(c[0], r, v, c[1 + NT : 1 + NT + NX]):
        out.append(d * gamma)
    return out


# ----------------------------------------------------------------------------
# Whole-network model for the plain-MLP family (fourier / feedforward / siren)
# ----------------------------------------------------------------------------
def mlp_program(spec, sd: Mapping[str, Tensor]) -> Dict:
    """Canonical layer program the kernels execute: encoding + hidden Linear/act layers + output Linear."""
    a = spec.architecture
    if a == "fourier":
        n = spec.num_layers
        hidden = [(sd[f"model.layers.{i}.weight"], sd[f"model.layers.{i}.bias"], spec.activation, 0.0,
                   f"model.layers.{i}") for i in range(n - 1)]
        return {"enc": "fourier", "B": sd["model.fourier.B"], "hidden": hidden,
                "out": (sd[f"model.layers.{n - 1}.weight"], sd[f"model.layers.{n - 1}.bias"], f"model.layers.{n - 1}")}
    if a == "feedforward":
        assert not spec.layer_norm
        hs = spec.dims()
        hidden = [(sd[f"model.layers.{2 * i}.weight"], sd[f"model.layers.{2 * i}.bias"], spec.activation, 0.0,
                   f"model.layers.{2 * i}") for i in range(len(hs))]
        k = 2 * len(hs)
        return {"enc": "linear", "hidden": hidden,
                "out": (sd[f"model.layers.{k}.weight"], sd[f"model.layers.{k}.bias"], f"model.layers.{k}")}
    if a == "siren":
        hs = spec.dims()
        hidden = [(sd[f"model.layers.{i}.linear.weight"], sd[f"model.layers.{i}.linear.bias"], "sin", spec.omega_0,
                   f"model.layers.{i}.linear") for i in range(len(hs))]
        k = len(hs)
        return {"enc": "linear", "hidden": hidden,
                "out": (sd[f"model.layers.{k}.weight"], sd[f"model.layers.{k}.bias"], f"model.layers.{k}")}
    raise ValueError(a)


def input_streams(inp: Tensor, NT: int, NX: int) -> List[Tensor]:
    """Jets of the identity input map: value = inp, d/dt = e_time (last column), d/dx = e_0."""
    N, din = inp.shape
    K = 1 + NT + NX
    s = [inp] + [torch.zeros_like(inp) for _ in range(K - 1)]
    if NT >= 1:
        s[1][:, din - 1] = 1.0
    if NX >= 1:
        s[1 + NT][:, 0] = 1.0
    return s


def mlp_jets_forward(prog: Dict, inp: Tensor, NT: int, NX: int):
    """Returns (jets [K tensors (N,1)], tape) where tape holds what the reverse sweep re-reads."""
    K = 1 + NT + NX
    a = input_streams(inp, NT, NX)
    tape = {"inp": inp, "z": [], "a_in": []}
    if prog["enc"] == "fourier":
        z = [s @ prog["B"] for s in a]  # linear map of the input jets
        sin_j = act_fwd("sin", 1.0, z, NT, NX)
        cos_z = [z[0] + math.pi / 2] + z[1:]  # cos(z) = sin(z + pi/2): same jets
        cos_j = act_fwd("sin", 1.0, cos_z, NT, NX)
        a = [torch.cat([s, c], -1) for s, c in zip(sin_j, cos_j)]
    for W, b, act, par, _ in prog["hidden"]:
        tape["a_in"].append(a)
        z = [a[0] @ W.T + b] + [a[s] @ W.T for s in range(1, K)]
        tape["z"].append(z)
        a = act_fwd(act, par, z, NT, NX)
    Wo, bo, _ = prog["out"]
    tape["a_last"] = a
    u = [a[0] @ Wo.T + bo] + [a[s] @ Wo.T for s in range(1, K)]
    return u, tape


def mlp_jets_backward(prog: Dict, tape: Dict, ubar: List[Tensor], NT: int, NX: int) -> Dict[str, Tensor]:
    """Reverse sweep given cotangents of the K output jets; returns {param name: grad}."""
    K = 1 + NT + NX
    g: Dict[str, Tensor] = {}
    Wo, bo, no = prog["out"]
    a = tape["a_last"]
    g[no + ".weight"] = sum(ubar[s].T @ a[s] for s in range(K))
    g[no + ".bias"] = ubar[0].sum(0)
    abar = [ubar[s] @ Wo for s in range(K)]
    for li in range(len(prog["hidden"]) - 1, -1, -1):
        W, b, act, par, nm = prog["hidden"][li]
        zb = act_bwd(act, par, tape["z"][li], abar, NT, NX)
        a_in = tape["a_in"][li]
        g[nm + ".weight"] = sum(zb[s].T @ a_in[s] for s in range(K))
        g[nm + ".bias"] = zb[0].sum(0)
        abar = [zb[s] @ W for s in range(K)]
    return g


# ----------------------------------------------------------------------------
# PDE epilogues (residual and its cotangents w.r.t. the jets)
# ----------------------------------------------------------------------------
def pde_streams(name: str, dimension: int = 1) -> Tuple[int, int]:
    """(NT, NX) the as-reference residual of each PDE consumes (SURVEY §0.3 quirks included)."""
    if dimension > 1:
        return {"wave": (2, 0), "pendulum": (2, 0)}.get(name, (1, 0))
    return {"heat": (1, 1), "burgers": (1, 2), "allen_cahn": (1, 2), "kdv": (1, 3), "cahn_hilliard": (1, 4),
            "wave": (2, 2), "convection": (1, 1), "black_scholes": (1, 2), "pendulum": (2, 0)}[name]


def pde_residual(name: str, p: Mapping, j: List[Tensor], x0: Tensor, NT: int, NX: int, dimension: int = 1):
    """Returns (r, [dr/dj_s]) per point.  j = jets in stream order; x0 = first spatial coordinate."""
    u = j[0]
    T = lambda k: j[k]  # noqa: E731  time derivative order k
    X = lambda k: j[NT + k]  # noqa: E731  space derivative order k
    zero = torch.zeros_like(u)
    d = [zero.clone() for _ in j]
    one = torch.ones_like(u)
    if dimension > 1:
        if name == "wave":
            d[2] = one
            return T(2), d
        if name == "pendulum":
            gl = p.get("g", 9.81) / p.get("L", 1.0)
            d[0], d[2] = gl * torch.cos(u), one
            return T(2) + gl * torch.sin(u), d
        if name == "allen_cahn":
            d[0], d[1] = 3 * u * u - 1, one
            return T(1) - u + u**3, d
        d[1] = one  # every other >=2-D residual keeps only its u_t term (+ nothing that survives)
        if name == "burgers" or name == "kdv" or name == "heat" or name == "cahn_hilliard" or name == "convection":
            return T(1), d
        raise ValueError(name)
    if name == "burgers":
        nu = p.get("nu", 0.01)
        d[0], d[1], d[NT + 1], d[NT + 2] = X(1), one, u, -nu * one
        return T(1) + u * X(1) - nu * X(2), d
    if name == "heat":
        al = p["alpha"]
        d[1], d[NT + 1] = one, -al * one
        return T(1) - al * X(1), d
    if name == "allen_cahn":
        e2 = p.get("epsilon", 0.1) ** 2
        d[0], d[1], d[NT + 2] = 3 * u * u - 1, one, -e2 * one
        return T(1) - e2 * X(2) - u + u**3, d
    if name == "kdv":
        d[0], d[1], d[NT + 1], d[NT + 3] = 6 * X(1), one, 6 * u, one
        return T(1) + 6 * u * X(1) + X(3), d
    if name == "cahn_hilliard":
        e2 = p.get("epsilon", 0.1) ** 2
        m = ((u >= -10.0) & (u <= 10.0)).to(u.dtype)
        c = torch.clamp(u, -10.0, 10.0)
        r = T(1) + e2 * X(4) - m * (6 * c * X(1) ** 2 + (3 * c * c - 1) * X(2))
        d[0] = -m * (6 * X(1) ** 2 + 6 * c * X(2))
        d[1] = one
        d[NT + 1] = -m * 12 * c * X(1)
        d[NT + 2] = -m * (3 * c * c - 1)
        d[NT + 4] = e2 * one
        return r, d
    if name == "wave":
        c2 = p.get("c", 1.0) ** 2
        d[2], d[NT + 2] = one, -c2 * one
        return T(2) - c2 * X(2), d
    if name == "convection":
        v = p.get("velocity", [1.0])
        v = v[0] if isinstance(v, (list, tuple)) else v
        d[1], d[NT + 1] = one, v * one
        return T(1) + v * X(1), d
    if name == "black_scholes":
        sg, rr = p.get("sigma", 0.2), p.get("r", 0.05)
        d[0], d[1], d[NT + 1], d[NT + 2] = -rr * one, one, rr * x0, 0.5 * sg**2 * x0**2
        return T(1) + 0.5 * sg**2 * x0**2 * X(2) + rr * x0 * X(1) - rr * u, d
    if name == "pendulum":
        gl = p.get("g", 9.81) / p.get("L", 1.0)
        d[0], d[2] = gl * torch.cos(u), one
        return T(2) + gl * torch.sin(u), d
    raise ValueError(name)
