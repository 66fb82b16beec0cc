"""Plain PyTorch-CPU restatement of pinnrl's collocation-point hot path.

TEST INFRASTRUCTURE — see `oracle/__init__.py`.  Every function cites the
reference file:line (relative to `/root/reference/`) it follows.  The op
sequence is kept identical to the reference (two forwards for Burgers/KdV,
chained `torch.autograd.grad(create_graph=True)`, then `backward`), so that this
module timed on host cores IS the "reference CPU path" cost (`cpu_baseline`
kind "port").

Networks are functional: they take a `state_dict`-style mapping with the
reference's key names (`model.fourier.B`, `model.layers.0.weight`, ...), so the
same function evaluates reference checkpoints, golden fixtures and the product's
own parameters.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# Specs (plain data; mirror the fields the reference reads)
# ----------------------------------------------------------------------------
@dataclass
class ArchSpec:
    """Fields of `ModelConfig` the networks read (pinnrl/config/__init__.py:172-253)."""

    architecture: str = "fourier"
    input_dim: int = 2
    hidden_dim: int = 128
    num_layers: int = 4
    output_dim: int = 1
    activation: str = "tanh"
    mapping_size: int = 32  # config/__init__.py:212
    scale: float = 10.0  # config/__init__.py:213
    omega_0: float = 30.0
    num_blocks: Optional[int] = None
    num_heads: int = 4
    layer_norm: bool = False
    hidden_dims: Optional[List[int]] = None

    def dims(self) -> List[int]:
        # ModelConfig.__init__: hidden_dims = [hidden_dim] * num_layers (config/__init__.py:241)
        return list(self.hidden_dims) if self.hidden_dims else [self.hidden_dim] * self.num_layers


@dataclass
class PdeSpec:
    """Fields of `PDEConfig` the residual/sampling path reads (pinnrl/pdes/pde_base.py:22-47)."""

    name: str = "burgers"
    dimension: int = 1
    domain: Sequence[Tuple[float, float]] = ((-1.0, 1.0),)
    time_domain: Tuple[float, float] = (0.0, 1.0)
    parameters: Dict[str, Any] = field(default_factory=dict)
    boundary_conditions: Dict[str, Dict[str, Any]] = field(default_factory=dict)
    initial_condition: Dict[str, Any] = field(default_factory=dict)
    loss_weights: Optional[Dict[str, float]] = None
    loss_function: str = "mse"
    huber_delta: float = 1.0


# ----------------------------------------------------------------------------
# Parameter initialisation in the reference's construction order
# ----------------------------------------------------------------------------
def _linear(sd: Dict[str, Tensor], prefix: str, fan_in: int, fan_out: int) -> nn.Linear:
    lin = nn.Linear(fan_in, fan_out)  # torch default init; consumes the global RNG like the reference
    sd[prefix + ".weight"] = lin.weight.detach().clone()
    sd[prefix + ".bias"] = lin.bias.detach().clone()
    return lin


def _layernorm(sd: Dict[str, Tensor], prefix: str, dim: int) -> None:
    sd[prefix + ".weight"] = torch.ones(dim)
    sd[prefix + ".bias"] = torch.zeros(dim)


def init_state_dict(spec: ArchSpec, seed: Optional[int] = None) -> Dict[str, Tensor]:
    """theta_0 with the reference's RNG consumption order, keyed like `PINNModel.state_dict()`.

    Follows PINNModel.__init__ (neural_networks/__init__.py:69-142) and the
    constructors it dispatches to; must be called under the same
    `torch.manual_seed` the reference model was built under.
    """
    if seed is not None:
        torch.manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    a = spec.architecture
    if a == "fourier":
        # fourier.py:45 — B drawn BEFORE the Linear layers; fourier.py:93-105
        sd["model.fourier.B"] = torch.randn(spec.input_dim, spec.mapping_size) * spec.scale
        prev = 2 * spec.mapping_size
        for i in range(spec.num_layers - 1):
            _linear(sd, f"model.layers.{i}", prev, spec.hidden_dim)
            prev = spec.hidden_dim
        _linear(sd, f"model.layers.{spec.num_layers - 1}", prev, spec.output_dim)
    elif a == "feedforward":
        # feedforward.py:37-53 — Sequential indices advance by 2 (Linear, act) or 3 (+LayerNorm)
        idx, prev = 0, spec.input_dim
        for h in spec.dims():
            _linear(sd, f"model.layers.{idx}", prev, h)
            idx += 1
            if spec.layer_norm:
                _layernorm(sd, f"model.layers.{idx}", h)
                idx += 1
            idx += 1  # activation module (no parameters); dropout=0.0 adds no module
            prev = h
        _linear(sd, f"model.layers.{idx}", prev, spec.output_dim)
    elif a == "siren":
        # siren.py:25-34,69-75 — nn.Linear init first, then the SIREN uniform overwrite of the weight
        prev = spec.input_dim
        hs = spec.dims()
        for i, h in enumerate(hs):
            _linear(sd, f"model.layers.{i}.linear", prev, h)
            bound = math.sqrt(6 / prev) / spec.omega_0
            sd[f"model.layers.{i}.linear.weight"] = torch.empty(h, prev).uniform_(-bound, bound)
            prev = h
        _linear(sd, f"model.layers.{len(hs)}", prev, spec.output_dim)
    elif a == "resnet":
        # resnet.py:114-127, 43-52; num_blocks falls back to num_layers (__init__.py:110-113)
        nb = spec.num_blocks if spec.num_blocks is not None else spec.num_layers
        H = spec.hidden_dim
        _linear(sd, "model.input_layer", spec.input_dim, H)
        for b in range(nb):
            _linear(sd, f"model.blocks.{b}.layers.0", H, H)
            _layernorm(sd, f"model.blocks.{b}.layers.1", H)
            _linear(sd, f"model.blocks.{b}.layers.4", H, H)
            _layernorm(sd, f"model.blocks.{b}.layers.5", H)
        _linear(sd, "model.output_layer", H, spec.output_dim)
    elif a == "attention":
        # attention.py:136-156 build, then apply(_init_weights) re-draws every Linear N(0, 0.02), bias 0
        H = spec.hidden_dim
        names: List[Tuple[str, int, int]] = [("model.input_proj", spec.input_dim, H)]
        for l in range(spec.num_layers):
            for nm in ("query", "key", "value", "proj"):
                names.append((f"model.layers.{l}.0.{nm}", H, H))
            names.append((f"model.layers.{l}.0.layer_norm", 0, H))  # fan_in 0 marks a LayerNorm
            names.append((f"model.layers.{l}.1.net.0", H, 4 * H))
            names.append((f"model.layers.{l}.1.net.3", 4 * H, H))
            names.append((f"model.layers.{l}.1.layer_norm", 0, H))
        names.append(("model.output_proj", H, spec.output_dim))
        for nm, fi, fo in names:  # construction pass (Linear draws consume RNG, values discarded below)
            if fi:
                _linear(sd, nm, fi, fo)
            else:
                _layernorm(sd, nm, fo)
        for nm, fi, fo in names:  # attention.py:158-163 via nn.Module.apply (children in registration order)
            if fi:
                sd[nm + ".weight"] = torch.empty(fo, fi).normal_(mean=0.0, std=0.02)
                sd[nm + ".bias"] = torch.zeros(fo)
    else:
        raise ValueError(f"oracle: architecture '{a}' is outside the hot-path scope")
    return sd


# ----------------------------------------------------------------------------
# Network forward (functional)
# ----------------------------------------------------------------------------
def _act(name: str):
    # base_network.py:91-104
    if name == "relu":
        return F.relu
    if name == "leaky_relu":
        return F.leaky_relu
    if name == "tanh":
        return torch.tanh
    if name == "sigmoid":
        return torch.sigmoid
    if name == "gelu":
        return F.gelu
    raise ValueError(f"Unsupported activation: {name}")


def composite_layer_norm(x: Tensor, shape, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm written with plain mean / rsqrt ops: the same function as `F.layer_norm` (nn.LayerNorm,
    resnet.py:43-52, feedforward.py:43-45, attention.py:36,94), but differentiated by autograd op by op.

    torch 2.10's fused `layer_norm` returns a wrong THIRD derivative (double-backward of its backward), so for
    networks with a LayerNorm the reference's d(loss)/d(theta) is not the derivative of its own loss whenever the
    residual holds a second (or higher) input derivative.  `layer_norm="composite"` is the exact-derivative
    checker: equal to the fused path up to second input derivatives (tests/test_oracle_golden.py) and equal to
    finite differences for the parameter gradient."""
    mu = x.mean(-1, keepdim=True)
    c = x - mu
    var = (c * c).mean(-1, keepdim=True)
    return c * torch.rsqrt(var + eps) * weight + bias


def network_forward(spec: ArchSpec, sd: Mapping[str, Tensor], inp: Tensor, layer_norm: str = "fused") -> Tensor:
    """`PINNModel.forward` (neural_networks/__init__.py:144-154) on a state_dict.

    `layer_norm="fused"` is the reference's op (`F.layer_norm`); `"composite"` evaluates the same LayerNorm with
    plain ops (see `composite_layer_norm`)."""
    a = spec.architecture
    x = inp
    LN = F.layer_norm if layer_norm == "fused" else composite_layer_norm
    if a == "fourier":
        # fourier.py:12-16 (x @ B, cat[sin, cos]); fourier.py:120-124
        proj = x @ sd["model.fourier.B"]
        x = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
        act = _act(spec.activation)
        n = spec.num_layers
        for i in range(n - 1):
            x = act(F.linear(x, sd[f"model.layers.{i}.weight"], sd[f"model.layers.{i}.bias"]))
        return F.linear(x, sd[f"model.layers.{n - 1}.weight"], sd[f"model.layers.{n - 1}.bias"])
    if a == "feedforward":
        # feedforward.py:37-53,59-73
        act = _act(spec.activation)
        idx = 0
        for h in spec.dims():
            x = F.linear(x, sd[f"model.layers.{idx}.weight"], sd[f"model.layers.{idx}.bias"])
            idx += 1
            if spec.layer_norm:
                x = LN(x, (h,), sd[f"model.layers.{idx}.weight"], sd[f"model.layers.{idx}.bias"])
                idx += 1
            x = act(x)
            idx += 1
        return F.linear(x, sd[f"model.layers.{idx}.weight"], sd[f"model.layers.{idx}.bias"])
    if a == "siren":
        # siren.py:36-46,77-90
        n = len(spec.dims())
        for i in range(n):
            z = F.linear(x, sd[f"model.layers.{i}.linear.weight"], sd[f"model.layers.{i}.linear.bias"])
            x = torch.sin(spec.omega_0 * z)
        return F.linear(x, sd[f"model.layers.{n}.weight"], sd[f"model.layers.{n}.bias"])
    if a == "resnet":
        # resnet.py:128-142 and block forward :54-65 (Dropout p=0 is the identity)
        act = _act(spec.activation)
        nb = spec.num_blocks if spec.num_blocks is not None else spec.num_layers
        H = spec.hidden_dim
        x = act(F.linear(x, sd["model.input_layer.weight"], sd["model.input_layer.bias"]))
        for b in range(nb):
            p = f"model.blocks.{b}.layers."
            y = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
            y = LN(y, (H,), sd[p + "1.weight"], sd[p + "1.bias"])
            y = act(y)
            y = F.linear(y, sd[p + "4.weight"], sd[p + "4.bias"])
            y = LN(y, (H,), sd[p + "5.weight"], sd[p + "5.bias"])
            x = act(x + y)
        return F.linear(x, sd["model.output_layer.weight"], sd["model.output_layer.bias"])
    if a == "attention":
        # attention.py:165-183; SelfAttention.forward :39-72 with a length-1 sequence
        act = _act(spec.activation)
        H, nh = spec.hidden_dim, spec.num_heads
        hd = H // nh
        x = act(F.linear(x, sd["model.input_proj.weight"], sd["model.input_proj.bias"]))
        for l in range(spec.num_layers):
            p = f"model.layers.{l}.0."
            res = x
            bsz = x.shape[0]
            xs = x.unsqueeze(1)
            q = F.linear(xs, sd[p + "query.weight"], sd[p + "query.bias"]).view(bsz, 1, nh, hd).transpose(1, 2)
            k = F.linear(xs, sd[p + "key.weight"], sd[p + "key.bias"]).view(bsz, 1, nh, hd).transpose(1, 2)
            v = F.linear(xs, sd[p + "value.weight"], sd[p + "value.bias"]).view(bsz, 1, nh, hd).transpose(1, 2)
            scores = torch.matmul(q, k.transpose(-2, -1)) * (hd**-0.5)
            attn = F.softmax(scores, dim=-1)
            out = torch.matmul(attn, v).transpose(1, 2).contiguous().view(bsz, 1, H)
            out = F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"]).squeeze(1)
            x = LN(out + res, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
            p = f"model.layers.{l}.1."
            y = F.linear(x, sd[p + "net.0.weight"], sd[p + "net.0.bias"])
            y = F.gelu(y)  # attention.py:90 — nn.GELU() regardless of config.activation
            y = F.linear(y, sd[p + "net.3.weight"], sd[p + "net.3.bias"])
            x = LN(x + y, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
        return F.linear(x, sd["model.output_proj.weight"], sd["model.output_proj.bias"])
    raise ValueError(f"oracle: architecture '{a}' is outside the hot-path scope")


# ----------------------------------------------------------------------------
# compute_derivatives / residuals
# ----------------------------------------------------------------------------
def _grad(y: Tensor, wrt: Tensor) -> Optional[Tensor]:
    return torch.autograd.grad(
        y, wrt, grad_outputs=torch.ones_like(y), create_graph=True, allow_unused=True, retain_graph=True
    )[0]


def compute_derivatives(
    model_fn,
    x: Tensor,
    t: Tensor,
    temporal_derivatives: Optional[Sequence[int]] = None,
    spatial_derivatives: Optional[Sequence[int]] = None,
    dimension: int = 1,
) -> Dict[str, Tensor]:
    """`PDEBase.compute_derivatives` (pdes/pde_base.py:590-794).

    Keeps the reference's chaining rule: the key is the ORDER REQUESTED, the
    value is however many `autograd.grad` calls were actually chained (so
    `spatial_derivatives=[2]` alone yields a FIRST derivative under "dx2").
    """
    if temporal_derivatives and max(temporal_derivatives) > 2:  # pde_base.py:614-619
        raise ValueError(
            f"Temporal derivative order {max(temporal_derivatives)} is not supported. Maximum order is 2."
        )
    if spatial_derivatives and max(spatial_derivatives) > 4:  # pde_base.py:621-627
        raise ValueError(
            f"Spatial derivative order {max(spatial_derivatives)} is not supported. Maximum order is 4."
        )
    x = x.detach().requires_grad_(True)  # pde_base.py:630-631
    t = t.detach().requires_grad_(True)
    u = model_fn(torch.cat([x, t], dim=1))  # pde_base.py:640-641
    d: Dict[str, Tensor] = {}
    if temporal_derivatives:  # pde_base.py:651-689
        prev = u
        for i in sorted(temporal_derivatives):
            if i == 0:
                continue
            g = _grad(u if i == 1 else prev, t)
            if g is None:
                g = torch.zeros_like(u)
            g = g.requires_grad_(True)  # pde_base.py:671,688 (turns a zeros fallback into a leaf)
            d["dt" if i == 1 else f"dt{i}"] = g
            prev = g
    if spatial_derivatives:
        if dimension == 1:  # pde_base.py:693-732
            prev = u
            for i in sorted(spatial_derivatives):
                if i == 0:
                    continue
                g = _grad(u if i == 1 else prev, x)
                if g is None:
                    g = torch.zeros_like(u)
                g = g.requires_grad_(True)  # pde_base.py:713,731
                d["dx" if i == 1 else f"dx{i}"] = g
                prev = g
        else:  # pde_base.py:733-779 — gradient w.r.t. a fresh slice is unused -> zeros
            for dim in range(dimension):
                name = f"x{dim + 1}"
                for order in sorted(spatial_derivatives):
                    if order == 0:
                        continue
                    prev = u
                    for i in range(1, order + 1):
                        g = _grad(u if i == 1 else prev, x[:, dim : dim + 1])
                        if g is None:
                            g = torch.zeros_like(u)
                        g = g.requires_grad_(True)  # pde_base.py:756,776
                        d[f"d{name * i}"] = g
                        prev = g
    if spatial_derivatives and 2 in spatial_derivatives:  # pde_base.py:781-792
        if dimension == 1:
            d["laplacian"] = d["dx2"]
        else:
            lap = torch.zeros_like(u)
            for dim in range(dimension):
                lap = lap + d[f"dx{dim + 1}x{dim + 1}"]
            d["laplacian"] = lap
    d["_x"], d["_t"] = x, t  # not in the reference dict; lets callers reuse the detached leaves
    return d


def _inline_chain(model_fn, x: Tensor, t: Tensor, dimension: int):
    """The hand-inlined autograd chain of allen_cahn.py:55-108 / cahn_hilliard.py:55-109 / wave_equation.py:53-116."""
    u = model_fn(torch.cat([x, t], dim=1))
    u_t = _grad(u, t)
    if u_t is None:
        u_t = torch.zeros_like(u)
    if dimension == 1:
        u_x = _grad(u, x)
        if u_x is None:
            u_x = torch.zeros_like(u)
        u_xx = _grad(u_x, x)
        if u_xx is None:
            u_xx = torch.zeros_like(u)
        lap = u_xx
    else:
        lap = torch.zeros_like(u)
        for dim in range(dimension):
            u_x = _grad(u, x[:, dim : dim + 1])  # fresh slice: not in u's graph -> None
            if u_x is not None:
                u_xx = _grad(u_x, x[:, dim : dim + 1])
                if u_xx is not None:
                    lap = lap + u_xx
    return u, u_t, lap


def compute_residual(pde: PdeSpec, model_fn, x: Tensor, t: Tensor) -> Tensor:
    """`XxxEquation.compute_residual(model, x, t)` for every PDE in pinnrl/pdes/."""
    p = pde.parameters
    dim = pde.dimension
    name = pde.name
    if name == "burgers":  # burgers_equation.py:40-75
        d = compute_derivatives(model_fn, x, t, temporal_derivatives=[1], spatial_derivatives=[1, 2], dimension=dim)
        u = model_fn(torch.cat([x, t], dim=1))  # second forward, burgers_equation.py:64
        nu = p.get("nu", 0.01)
        diffusion = nu * d["laplacian"]
        if dim == 1:
            conv = u * d["dx"]
        else:
            conv = torch.zeros_like(u)
            for k in range(dim):
                conv = conv + u * d[f"dx{k + 1}"]
        return d["dt"] + conv - diffusion
    if name == "heat":  # heat_equation.py:54-110 — asks for spatial [2] only: "laplacian" is u_x (quirk)
        d = compute_derivatives(model_fn, x, t, temporal_derivatives=[1], spatial_derivatives=[2], dimension=dim)
        return d["dt"] - p["alpha"] * d["laplacian"]
    if name == "allen_cahn":  # allen_cahn.py:39-111 (no detach; in-place requires_grad_)
        x = x.requires_grad_(True)
        t = t.requires_grad_(True)
        u, u_t, lap = _inline_chain(model_fn, x, t, dim)
        eps = p.get("epsilon", 0.1)
        return u_t - eps**2 * lap - u + u**3
    if name == "kdv":  # kdv_equation.py:38-92
        d = compute_derivatives(model_fn, x, t, temporal_derivatives=[1], spatial_derivatives=[1, 2, 3], dimension=dim)
        u = model_fn(torch.cat([x, t], dim=1))
        if dim == 1:
            return d["dt"] + 6 * u * d["dx"] + d["dx3"]
        r = d["dt"]
        for k in range(dim):
            nm = f"x{k + 1}"
            r = r + 6 * u * d[f"d{nm}"] + d[f"d{nm * 3}"]
        return r
    if name == "cahn_hilliard":  # cahn_hilliard.py:39-160
        x = x.detach().requires_grad_(True)
        t = t.detach().requires_grad_(True)
        u, u_t, lap = _inline_chain(model_fn, x, t, dim)
        eps = p.get("epsilon", 0.1)
        uc = torch.clamp(u, -10.0, 10.0)
        mu = -(eps**2) * lap + uc**3 - uc
        if dim == 1:
            mu_x = _grad(mu, x)
            if mu_x is None:
                mu_x = torch.zeros_like(mu)
            mu_xx = _grad(mu_x, x)
            if mu_xx is None:
                mu_xx = torch.zeros_like(mu)
            lap_mu = mu_xx
        else:
            lap_mu = torch.zeros_like(mu)
            for k in range(dim):
                mu_x = _grad(mu, x[:, k : k + 1])
                if mu_x is not None:
                    mu_xx = _grad(mu_x, x[:, k : k + 1])
                    if mu_xx is not None:
                        lap_mu = lap_mu + mu_xx
        return u_t - lap_mu
    if name == "wave":  # wave_equation.py:38-119
        x = x.requires_grad_(True)
        t = t.requires_grad_(True)
        u, u_t, lap = _inline_chain(model_fn, x, t, dim)
        u_tt = _grad(u_t, t)
        if u_tt is None:
            u_tt = torch.zeros_like(u)
        c = p.get("c", 1.0)
        return u_tt - c**2 * lap
    if name == "convection":  # convection_equation.py:43-78
        if dim > 1:  # :66-76 differentiates u w.r.t. x[:, d:d+1], a slice that is not part of u's graph: torch raises this
            raise RuntimeError("One of the differentiated Tensors appears to not have been used in the graph. "
                               "Set allow_unused=True if this is the desired behavior.")
        x = x.detach().requires_grad_(True)
        t = t.detach().requires_grad_(True)
        u = model_fn(torch.cat([x, t], dim=1))
        u_t = torch.autograd.grad(u, t, grad_outputs=torch.ones_like(u), create_graph=True)[0]
        vel = p.get("velocity", [1.0])
        if not isinstance(vel, (list, tuple)):
            vel = [vel]
        u_x = torch.autograd.grad(u, x, grad_outputs=torch.ones_like(u), create_graph=True)[0]
        return u_t + vel[0] * u_x
    if name == "black_scholes":  # black_scholes.py:44-93
        d = compute_derivatives(model_fn, x, t, temporal_derivatives=[1], spatial_derivatives=[1, 2], dimension=dim)
        xd = d["_x"]
        V = model_fn(torch.cat([xd, d["_t"]], dim=1))
        sigma, r = p.get("sigma", 0.2), p.get("r", 0.05)
        if dim > 1:  # black_scholes.py:84-91: first-dimension derivatives (zeros: fresh-slice quirk) broadcast over x
            return (d["dt"] + 0.5 * sigma**2 * torch.sum(xd**2 * d["dx1x1"], dim=1, keepdim=True)
                    + r * torch.sum(xd * d["dx1"], dim=1, keepdim=True) - r * V)
        return d["dt"] + 0.5 * sigma**2 * xd**2 * d["dx2"] + r * xd * d["dx"] - r * V
    if name == "pendulum":  # pendulum_equation.py:51-94
        d = compute_derivatives(model_fn, x, t, temporal_derivatives=[1, 2], spatial_derivatives=set(), dimension=dim)
        u = model_fn(torch.cat([d["_x"], d["_t"]], dim=1))
        g, L = p.get("g", 9.81), p.get("L", 1.0)
        return d["dt2"] + (g / L) * torch.sin(u)
    raise ValueError(f"oracle: unknown pde '{name}'")


def apply_loss_fn(error: Tensor, name: str = "mse", huber_delta: float = 1.0) -> Tensor:
    """`PDEBase._apply_loss_fn` (pdes/pde_base.py:309-326)."""
    if name == "mae":
        return torch.mean(torch.abs(error))
    if name == "huber":
        return F.huber_loss(error, torch.zeros_like(error), reduction="mean", delta=huber_delta)
    return torch.mean(error**2)


def residual_loss_and_grad(
    pde: PdeSpec, spec: ArchSpec, sd: Mapping[str, Tensor], x: Tensor, t: Tensor, layer_norm: str = "fused"
) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """The metric's unit of work: r = compute_residual; L = mean(r^2); L.backward().

    Returns (r detached, L detached, {param name: dL/dparam}).  `model.fourier.B`
    is a buffer in the reference (fourier.py:45) and gets no gradient.
    """
    params = {k: v.detach().clone().requires_grad_(k != "model.fourier.B") for k, v in sd.items()}
    r = compute_residual(pde, lambda inp: network_forward(spec, params, inp, layer_norm), x, t)
    L = apply_loss_fn(r, pde.loss_function, pde.huber_delta)
    names = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(L, [params[k] for k in names], allow_unused=True)
    out = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(names, grads)}
    return r.detach(), L.detach(), out


# ----------------------------------------------------------------------------
# Sampling and the non-residual loss terms (row T / A16)
# ----------------------------------------------------------------------------
def sample_uniform(pde: PdeSpec, num_points: int) -> Tuple[Tensor, Tensor]:
    """`PDEBase._sample_uniform` (pdes/pde_base.py:806-860) on CPU; consumes the global RNG identically."""
    if pde.dimension == 1:
        n_side = int(np.sqrt(num_points))  # pde_base.py:809
        (x0, x1), (t0, t1) = pde.domain[0], pde.time_domain
        xs = torch.linspace(x0, x1, n_side).reshape(-1, 1)
        ts = torch.linspace(t0, t1, n_side).reshape(-1, 1)
        X, T = torch.meshgrid(xs.squeeze(), ts.squeeze(), indexing="ij")
        x = X.reshape(-1, 1)
        t = T.reshape(-1, 1)
        x = x + torch.randn_like(x) * ((x1 - x0) * 0.01)
        t = t + torch.randn_like(t) * ((t1 - t0) * 0.01)
        return torch.clamp(x, x0, x1), torch.clamp(t, t0, t1)
    ppd = max(2, int(num_points ** (1 / (pde.dimension + 1))) + 1)  # pde_base.py:830
    grids = [torch.linspace(lo, hi, ppd) for lo, hi in pde.domain[: pde.dimension]]
    grids.append(torch.linspace(pde.time_domain[0], pde.time_domain[1], ppd))
    mesh = torch.meshgrid(*grids, indexing="ij")
    pts = torch.stack([g.reshape(-1) for g in mesh], dim=1)
    if len(pts) > num_points:
        pts = pts[torch.randperm(len(pts))[:num_points]]
    elif len(pts) < num_points:
        extra = torch.randint(0, len(pts), (num_points - len(pts),))
        pts = torch.cat([pts, pts[extra]], dim=0)
    pts = pts + torch.randn_like(pts) * 0.01
    for k in range(pde.dimension):
        pts[:, k] = torch.clamp(pts[:, k], pde.domain[k][0], pde.domain[k][1])
    pts[:, -1] = torch.clamp(pts[:, -1], pde.time_domain[0], pde.time_domain[1])
    return pts[:, : pde.dimension], pts[:, -1].reshape(-1, 1)


def _bc_fn(bc_type: str, params: Mapping[str, Any], dimension: int):
    """`PDEBase._create_boundary_condition` (pdes/pde_base.py:496-571), deterministic kinds only."""
    if bc_type in ("left", "right"):
        bc_type = "dirichlet"
    if bc_type in ("dirichlet", "neumann"):
        v = params.get("value", 0.0)
        return lambda x, t: torch.full_like(x[:, 0:1], v)
    if bc_type == "periodic":
        if dimension == 1:
            return lambda x, t: torch.sin(2 * torch.pi * x[:, 0:1])
        return lambda x, t: torch.sin(2 * torch.pi * torch.sum(x, dim=1, keepdim=True))
    if bc_type == "initial":
        k = params.get("type", "sine")
        if k in ("sine", "sin_exp_decay"):
            amp, fr = params.get("amplitude", 1.0), params.get("frequency", 1.0)
            return lambda x, t: amp * torch.sin(fr * torch.pi * x[:, 0:1])
        if k == "tanh":
            eps = params.get("epsilon", 0.1)
            return lambda x, t: torch.tanh(x[:, 0:1] / eps)
        if k == "gaussian":
            m, s = params.get("mean", 0.0), params.get("std", 0.1)
            return lambda x, t: torch.exp(-((x[:, 0:1] - m) ** 2) / (2 * s**2))
        if k == "fixed":
            v = params.get("value", 0.0)
            return lambda x, t: torch.full_like(x[:, 0:1], v)
        if k == "small_angle":
            v = params.get("initial_angle", 0.5)
            return lambda x, t: torch.full_like(x[:, 0:1], v)
        return lambda x, t: torch.zeros_like(x[:, 0:1])
    return lambda x, t: torch.zeros_like(x[:, 0:1])


def compute_loss_terms(pde: PdeSpec, model_fn, x: Tensor, t: Tensor, observations: Optional[Mapping[str, Tensor]] = None,
                       mode: str = "forward") -> Dict[str, Tensor]:
    """`PDEBase.compute_loss` (pdes/pde_base.py:1086-1235), fixed weights.

    Includes the reference's behaviour that the "initial" entry added by
    `_setup_boundary_conditions` (pde_base.py:483-486) is ALSO enforced on the
    200 boundary points (pde_base.py:1129-1132).

    `observations` ({"x", "t", "u"}) adds the data term of `_compute_data_loss` (pde_base.py:281-291); `mode` gates the
    total as pde_base.py:1187-1233 does: "data_only" drops the physics terms from the total (they are still returned),
    "inverse" / "data_only" / "data_augmented" force a positive data weight.  A trainable coefficient (inverse mode) is a
    tensor with requires_grad in `pde.parameters`.
    """
    residual = compute_residual(pde, model_fn, x, t)
    lf = lambda e: apply_loss_fn(e, pde.loss_function, pde.huber_delta)  # noqa: E731
    residual_loss = lf(residual)
    if pde.dimension == 1:
        xb = torch.tensor([pde.domain[0][0], pde.domain[0][1]], dtype=torch.float32).reshape(-1, 1)
    else:
        vals: List[float] = []
        for k in range(pde.dimension):
            vals.extend([pde.domain[k][0], pde.domain[k][1]])
        xb = torch.tensor(vals, dtype=torch.float32).reshape(-1, 1)
    tb = torch.linspace(pde.time_domain[0], pde.time_domain[1], 100).reshape(-1, 1)
    xb = xb.repeat_interleave(len(tb), dim=0)
    tb = tb.repeat(len(xb) // len(tb), 1)
    bcs = {k: _bc_fn(k, v, pde.dimension) for k, v in pde.boundary_conditions.items()}
    if "initial" not in bcs:
        bcs["initial"] = _bc_fn("initial", pde.initial_condition, pde.dimension)
    boundary_loss = torch.tensor(0.0)
    for fn in bcs.values():
        ub = model_fn(torch.cat([xb, tb], dim=1))
        boundary_loss = boundary_loss + lf(ub - fn(xb, tb))
    xi = torch.linspace(pde.domain[0][0], pde.domain[0][1], 100).reshape(-1, 1)
    ti = torch.zeros_like(xi)
    ui = model_fn(torch.cat([xi, ti], dim=1))
    initial_loss = lf(ui - bcs["initial"](xi, ti))
    lw = pde.loss_weights
    if lw:
        rw = lw.get("pde", lw.get("residual", 1.0))
        bw, iw = lw.get("boundary", 10.0), lw.get("initial", 10.0)
    else:
        rw, bw, iw = 1.0, 10.0, 10.0
    data_loss = torch.tensor(0.0)
    if observations:  # pde_base.py:281-291
        data_loss = lf(model_fn(torch.cat([observations["x"], observations["t"]], dim=1)) - observations["u"])
    data_weight = float(lw.get("data", 1.0)) if lw else 1.0  # _data_loss_weight, pde_base.py:328-336
    active = 0.0 if mode == "data_only" else 1.0
    if mode in ("inverse", "data_only", "data_augmented") and data_weight <= 0.0:
        data_weight = 1.0
    total = active * rw * residual_loss + active * bw * boundary_loss + active * iw * initial_loss + data_weight * data_loss
    return {"residual": residual_loss, "boundary": boundary_loss, "initial": initial_loss, "data": data_loss, "total": total}


def compute_loss_terms_heat(pde: PdeSpec, model_fn, x: Tensor, t: Tensor, num_boundary_points: Optional[int] = None,
                            num_initial_points: Optional[int] = None) -> Dict[str, Tensor]:
    """`HeatEquation.compute_loss` (pdes/heat_equation.py:375-623), 1-D, forward mode, fixed weights, no smoothness.

    Periodic BC enforced on u AND on du/dx at the two ends (boundary du/dx by autograd w.r.t. the boundary points,
    heat_equation.py:420-445), time points clustered in the first 1 % of the horizon, IC points clustered near the ends.
    """
    residual = compute_residual(pde, model_fn, x, t)
    lf = lambda e: apply_loss_fn(e, pde.loss_function, pde.huber_delta)  # noqa: E731
    residual_loss = lf(residual)
    nbp = num_boundary_points if num_boundary_points is not None else max(len(x) // 10, 10)
    t_max = pde.time_domain[1]
    t_early = t_max * 0.01
    n_early = max(nbp // 4, 1)
    tb = torch.cat([torch.linspace(0, t_early, n_early), torch.linspace(t_early, t_max, nbp - n_early)]).reshape(-1, 1)
    x_lo, x_hi = pde.domain[0]
    pl = torch.cat([torch.full((nbp, 1), x_lo), tb], dim=1).requires_grad_(True)
    pr = torch.cat([torch.full((nbp, 1), x_hi), tb], dim=1).requires_grad_(True)
    ul, ur = model_fn(pl), model_fn(pr)
    dl = torch.autograd.grad(ul, pl, grad_outputs=torch.ones_like(ul), create_graph=True)[0][:, 0:1]
    dr = torch.autograd.grad(ur, pr, grad_outputs=torch.ones_like(ur), create_graph=True)[0][:, 0:1]
    boundary_loss = torch.tensor(0.0) + lf(ul - ur) + lf(dl - dr)
    nip = num_initial_points if num_initial_points is not None else max(len(x) // 5, 10)
    xb = (x_hi - x_lo) * 0.1
    xi = torch.cat([torch.linspace(x_lo, x_lo + xb, nip // 4), torch.linspace(x_lo + xb, x_hi - xb, nip // 2),
                    torch.linspace(x_hi - xb, x_hi, nip // 4)]).reshape(-1, 1)
    ti = torch.zeros_like(xi)
    ui = model_fn(torch.cat([xi, ti], dim=1))
    ic = pde.initial_condition
    kind = ic.get("type", "sine")
    A, k = ic.get("amplitude", 1.0), ic.get("frequency", 2.0)
    wn = 2 * torch.pi * k / (x_hi - x_lo)  # heat_equation.py:214-262: the heat IC uses the wave number 2 pi k / L
    if kind == "sin_exp_decay":
        target = A * torch.sin(wn * xi) * torch.exp(-(pde.parameters["alpha"] * wn**2) * ti)
    else:
        target = A * torch.sin(wn * xi)
    initial_loss = lf(ui - target)
    lw = pde.loss_weights
    if lw:
        rw, bw, iw = lw.get("pde", lw.get("residual", 1.0)), lw.get("boundary", 10.0), lw.get("initial", 10.0)
    else:
        rw, bw, iw = 1.0, 10.0, 10.0
    total = rw * residual_loss + bw * boundary_loss + iw * initial_loss
    return {"residual": residual_loss, "boundary": boundary_loss, "initial": initial_loss, "total": total}


def rar_probabilities(pde: PdeSpec, model_fn, x_pool: Tensor, t_pool: Tensor) -> Tensor:
    """The sampling weights of `PDEBase._sample_residual_based` (pdes/pde_base.py:912-930) on a given candidate pool:
    |r| + 1e-8, normalised.  (The pool itself is `sample_uniform(4 N)`, the draw `torch.multinomial`.)"""
    residuals = compute_residual(pde, model_fn, x_pool.detach().requires_grad_(True), t_pool.detach().requires_grad_(True))
    mag = torch.abs(residuals.detach()).squeeze()
    probs = mag + 1e-8
    return probs / probs.sum()


def sample_stratified(pde: PdeSpec, num_points: int) -> Tuple[Tensor, Tensor]:
    """`PDEBase._sample_stratified` (pdes/pde_base.py:862-893) on CPU; consumes the global RNG identically."""
    bounds = [tuple(pde.domain[k]) for k in range(pde.dimension)] + [tuple(pde.time_domain)]
    samples = torch.zeros(num_points, len(bounds))
    for d, (lo, hi) in enumerate(bounds):
        bin_size = (hi - lo) / num_points
        offsets = torch.rand(num_points)
        indices = torch.arange(num_points, dtype=torch.float32)
        samples[:, d] = lo + (indices + offsets) * bin_size
        perm = torch.randperm(num_points)
        samples[:, d] = samples[perm, d]
    return samples[:, : pde.dimension], samples[:, -1].reshape(-1, 1)


# ----------------------------------------------------------------------------
# RL-driven ("adaptive") sampling: the DQN policy network and the epsilon-greedy scorer (row R / §8(f)2)
# ----------------------------------------------------------------------------
def dqn_init_state_dict(state_dim: int, action_dim: int, hidden_dim: int, num_layers: int = 3) -> Dict[str, Tensor]:
    """theta_0 of `DQNNetwork` (pinnrl/rl/rl_agent.py:15-76) with the reference's RNG consumption order:
    default nn.Linear draws while the Sequential is built, then `_init_weights` re-draws every Linear weight
    with xavier_normal_ (module order) and zeroes the biases."""
    sd: Dict[str, Tensor] = {}
    dims = [(state_dim, hidden_dim)] + [(hidden_dim, hidden_dim)] * (num_layers - 2)
    linear_keys = []
    for i, (fi, fo) in enumerate(dims):
        _linear(sd, f"layers.{i}.0", fi, fo)
        _layernorm(sd, f"layers.{i}.1", fo)
        linear_keys.append(f"layers.{i}.0")
    _linear(sd, f"layers.{num_layers - 1}", hidden_dim, action_dim)
    linear_keys.append(f"layers.{num_layers - 1}")
    for k in linear_keys:
        sd[k + ".weight"] = nn.init.xavier_normal_(torch.empty_like(sd[k + ".weight"]), gain=1.0)
        sd[k + ".bias"] = torch.zeros_like(sd[k + ".bias"])
    return sd


def dqn_forward(sd: Mapping[str, Tensor], x: Tensor, training: bool = True, dropout: float = 0.1) -> Tensor:
    """`DQNNetwork.forward` (rl_agent.py:78-88): (Linear, LayerNorm, ReLU, Dropout) x (L-1), Linear.
    The reference never calls `.eval()` on the policy network, so its Dropout(0.1) is ACTIVE while scoring."""
    n_hidden = sum(1 for k in sd if k.endswith(".0.weight"))
    for i in range(n_hidden):
        x = F.linear(x, sd[f"layers.{i}.0.weight"], sd[f"layers.{i}.0.bias"])
        x = F.layer_norm(x, (x.shape[-1],), sd[f"layers.{i}.1.weight"], sd[f"layers.{i}.1.bias"])
        x = F.dropout(F.relu(x), dropout, training)
    return F.linear(x, sd[f"layers.{n_hidden}.weight"], sd[f"layers.{n_hidden}.bias"])


@dataclass
class AgentState:
    """The fields of `RLAgent` the sampler touches (rl_agent.py:139-212): policy weights and the exploration rate."""

    policy: Dict[str, Tensor]
    epsilon: float = 1.0
    epsilon_end: float = 0.01
    epsilon_decay: float = 0.995
    dropout: float = 0.1
    training: bool = True


def make_agent(state_dim: int, action_dim: int, hidden_dim: int, **kw) -> AgentState:
    """`RLAgent.__init__` (rl_agent.py:139-212): builds policy_net, then target_net (a second init that only
    advances the RNG: its weights are overwritten by the policy's)."""
    policy = dqn_init_state_dict(state_dim, action_dim, hidden_dim)
    dqn_init_state_dict(state_dim, action_dim, hidden_dim)  # target_net: RNG consumption only
    return AgentState(policy=policy, **kw)


def select_action(agent: AgentState, state: Tensor) -> Tensor:
    """`RLAgent.select_action` (rl_agent.py:214-229).  Explore branch returns a (1, 1) tensor — with it the
    sampler's multinomial has ONE category and every point collapses onto grid cell 0."""
    if torch.rand(1).item() > agent.epsilon:
        with torch.no_grad():
            return dqn_forward(agent.policy, state, agent.training, agent.dropout).view(1, -1)
    return torch.rand(1, 1)


def sample_adaptive(pde: PdeSpec, num_points: int, agent: AgentState, history: List) -> Tuple[Tensor, Tensor]:
    """The `strategy == "adaptive"` branch of `PDEBase.generate_collocation_points` (pde_base.py:961-1073)."""
    G = min(100, max(10, int(np.sqrt(num_points))))
    grids = [torch.linspace(pde.domain[k][0], pde.domain[k][1], G) for k in range(pde.dimension)]
    grids.append(torch.linspace(pde.time_domain[0], pde.time_domain[1], G))
    mesh = torch.meshgrid(*grids, indexing="ij")
    points = torch.stack([g.flatten() for g in mesh], dim=1)
    with torch.no_grad():
        probs = torch.abs(select_action(agent, points))
        probs = probs / torch.sum(probs)
    idx = torch.multinomial(probs.flatten(), min(num_points, len(points)), replacement=True)
    sel = points[idx]
    if len(sel) < num_points:
        extra = torch.randint(0, len(sel), (num_points - len(sel),))
        sel = torch.cat([sel, sel[extra]], dim=0)
    noise_scale = min(0.01, min((pde.domain[k][1] - pde.domain[k][0]) / G for k in range(pde.dimension)),
                      (pde.time_domain[1] - pde.time_domain[0]) / G)
    sel = sel + torch.randn_like(sel) * noise_scale
    for k in range(pde.dimension):
        sel[:, k] = torch.clamp(sel[:, k], pde.domain[k][0], pde.domain[k][1])
    sel[:, -1] = torch.clamp(sel[:, -1], pde.time_domain[0], pde.time_domain[1])
    x = sel[:, 0].reshape(-1, 1) if pde.dimension == 1 else sel[:, : pde.dimension]
    t = sel[:, -1].reshape(-1, 1)
    history.append(sel.numpy().copy())
    if len(history) > 1:
        agent.epsilon = max(agent.epsilon_end, agent.epsilon * agent.epsilon_decay)  # rl_agent.py:557-566
    return x, t
