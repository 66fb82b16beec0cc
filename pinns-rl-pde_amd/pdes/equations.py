"""The nine PDE classes of pinnrl/pdes/*.py on top of the fused jet engine.

Each class keeps the reference's name, constructor, parameter properties, initial-condition
factory and exact solution (caller-side helpers), and declares its residual to the engine as
(`KIND`, coefficients): the fused PDE epilogue in csrc/jet_device.h evaluates it per point, with
the reference's behaviour INCLUDING its quirks (heat differentiates once, >= 2-D residuals lose
their spatial terms — SURVEY.md §0.3).  `_residual_from_jets` is the same formula as torch ops on
the kernel's jets; it is only used in inverse mode, where a coefficient is an `nn.Parameter` and
must stay in the autograd graph.
"""

from __future__ import annotations

import math
from typing import Any, Dict

import torch

from .pde_base import PDEBase, PDEConfig


def _as_float(v):
    return float(v.detach()) if isinstance(v, torch.Tensor) else float(v)


class BurgersEquation(PDEBase):
    """u_t + u u_x - nu u_xx  (burgers_equation.py:40-75)."""

    KIND = "burgers"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def nu(self):
        return self.get_parameter("nu", default=0.01)  # note: the YAML key `viscosity` is ignored upstream too

    def _coefficients(self):
        return (self.nu,)

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:
            return j[1]
        return j[1] + j[0] * j[nt + 1] - self.nu * j[nt + 2]

    def _create_boundary_condition(self, bc_type: str, params: Dict[str, Any]):  # burgers_equation.py:131-160
        if bc_type == "initial":
            kind = params.get("type", "sine")
            if kind == "sine":
                A, k = params.get("amplitude", -1.0), params.get("frequency", 1.0)
                if self.dimension == 1:
                    return lambda x, t: A * torch.sin(k * torch.pi * x)
                return lambda x, t: A * torch.prod(torch.sin(k * torch.pi * x), dim=1, keepdim=True)
            if kind == "tanh":
                eps = params.get("epsilon", 0.1)
                if self.dimension == 1:
                    return lambda x, t: torch.tanh((x - 0.5) / eps)
                return lambda x, t: torch.prod(torch.tanh((x - 0.5) / eps), dim=1, keepdim=True)
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # burgers_equation.py:77-129 (Cole-Hopf with phi_x in closed form)
        if not self.config.exact_solution:
            return None
        kind = self.config.exact_solution.get("type", "cole_hopf")
        nu0 = _as_float(self.nu)
        if kind == "cole_hopf":
            nu = self.config.exact_solution.get("viscosity", nu0)
            k = self.config.exact_solution.get("initial_frequency", 1.0)
            sol = torch.ones_like(x[:, 0:1])
            for d in range(self.dimension):
                xd = x[:, d : d + 1]
                decay = torch.exp(-nu * (k * torch.pi) ** 2 * t)
                phi = -torch.cos(k * torch.pi * xd) * decay
                phi_x = k * torch.pi * torch.sin(k * torch.pi * xd) * decay
                sol = sol * (-2 * nu * phi_x / phi)
            return sol
        if kind == "tanh":
            eps = self.config.exact_solution.get("epsilon", 0.1)
            sol = torch.ones_like(x[:, 0:1])
            for d in range(self.dimension):
                sol = sol * torch.tanh((x[:, d : d + 1] - 0.5 - nu0 * t) / eps)
            return sol
        raise ValueError(f"Unsupported exact solution type: {kind}")


class HeatEquation(PDEBase):
    """As-reference residual u_t - alpha * d["laplacian"], where "laplacian" is a FIRST x-derivative
    (heat_equation.py:54-110 requests spatial_derivatives=[2] only; pde_base.py:695-732 then chains once)."""

    KIND = "heat"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def alpha(self):
        return self.get_parameter("alpha", required=True)

    def _coefficients(self):
        return (self.alpha,)

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:
            return j[1]
        return j[1] - self.alpha * j[nt + 1]

    def validate(self, model, num_points: int = 5000) -> Dict[str, Any]:  # heat_equation.py:296-372
        """The heat class's own validation: the three error metrics of the base class, a finiteness check, the optional
        `physical_bounds` of the configuration, the periodic-boundary mismatch when a periodic condition is configured (at
        x = 0 and x = 1, as the reference hard-codes), `validation_passed` and `validation_messages`."""
        passed, messages, metrics = True, [], {}
        x, t = self.generate_collocation_points(num_points)
        u_pred = model(torch.cat([x, t], dim=1))
        if not bool(torch.isfinite(u_pred).all()):
            passed = False
            messages.append("Error: Solution contains NaN or Inf values")
        err = torch.abs(u_pred - self.exact_solution(x, t))
        metrics.update({"l2_error": torch.mean(err**2).item(), "max_error": torch.max(err).item(), "mean_error": torch.mean(err).item()})
        if hasattr(self.config, "physical_bounds"):
            lo = self.config.physical_bounds.get("min_temperature", float("-inf"))
            hi = self.config.physical_bounds.get("max_temperature", float("inf"))
            if bool(torch.any(u_pred < lo)) or bool(torch.any(u_pred > hi)):
                passed = False
                messages.append(f"Error: Solution violates physical temperature bounds [{lo}, {hi}]")
        if "periodic" in self.boundary_conditions:
            n = num_points // 10
            tb = torch.linspace(0, self.config.time_domain[1], n, device=self.device).reshape(-1, 1)
            u_l = model(torch.cat([torch.zeros(n, 1, device=self.device), tb], dim=1))
            u_r = model(torch.cat([torch.ones(n, 1, device=self.device), tb], dim=1))
            periodic_error = torch.mean((u_l - u_r) ** 2).item()
            metrics["periodic_bc_error"] = periodic_error
            if periodic_error > 1e-3:
                passed = False
                messages.append(f"Warning: Periodic boundary condition error ({periodic_error:.2e}) exceeds tolerance")
        metrics["validation_passed"] = passed
        metrics["validation_messages"] = messages
        return metrics

    def exact_solution_sine(self, x, t):  # heat_equation.py:197-212 ("legacy" form: wave number k pi, not 2 pi k / L)
        A = self.config.exact_solution.get("amplitude", 1.0)
        k = self.config.exact_solution.get("frequency", 2.0)
        decay = torch.exp(-self.alpha * (k * torch.pi) ** 2 * t)
        if self.dimension == 1:
            return A * decay * torch.sin(k * torch.pi * x)
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            sol = sol * (A * decay * torch.sin(k * torch.pi * x[:, d : d + 1]))
        return sol

    def _calculate_decay_rate(self, k: float):  # heat_equation.py:40-52
        L = self.config.domain[0][1] - self.config.domain[0][0]
        return self.alpha * (2 * torch.pi * k / L) ** 2

    def _create_boundary_condition(self, bc_type: str, params: Dict[str, Any]):  # heat_equation.py:214-300
        if bc_type == "initial":
            kind = params.get("type", "sine")
            A, k = params.get("amplitude", 1.0), params.get("frequency", 2.0)
            L = self.config.domain[0][1] - self.config.domain[0][0]
            wn = 2 * torch.pi * k / L
            if kind == "sin_exp_decay":
                dr = self._calculate_decay_rate(k)
                if self.dimension == 1:
                    return lambda x, t: A * torch.sin(wn * x) * torch.exp(-dr * t)

                def ic(x, t):
                    sol = torch.ones_like(x[:, 0:1])
                    for d in range(self.dimension):
                        Ld = self.config.domain[d][1] - self.config.domain[d][0]
                        sol = sol * torch.sin(2 * torch.pi * k / Ld * x[:, d : d + 1])
                    return A * sol * torch.exp(-dr * t)

                return ic
            if kind == "sine":
                if self.dimension == 1:
                    return lambda x, t: A * torch.sin(wn * x)
                return lambda x, t: A * torch.prod(torch.sin(wn * x), dim=1, keepdim=True)
        return super()._create_boundary_condition(bc_type, params)

    def _num_points(self, key: str, divisor: int, n: int) -> int:
        tr = self.config.training
        if tr is None:
            return max(n // divisor, 10)
        if isinstance(tr, dict):
            return tr.get(key, tr.get("num_collocation_points", n) // divisor)
        return getattr(tr, key, tr.num_collocation_points // divisor)

    def compute_loss(self, model, x, t, n_total=None, aux_scale: float = 1.0):
        """heat_equation.py:375-623 — periodic BC (u and du/dx match at the two ends, points clustered near t = 0),
        IC points clustered near the ends, optional finite-difference smoothness term.  The boundary du/dx that the
        reference obtains with `autograd.grad(create_graph=True)` is the x-stream of the jet kernel."""
        dev = self.device
        residual_loss = self._residual_loss(model, x, t, n_total)
        nbp = self._num_points("num_boundary_points", 10, len(x))
        t_max = self.config.time_domain[1]
        t_early = t_max * 0.01
        n_early = max(nbp // 4, 1)
        tb = torch.cat([torch.linspace(0, t_early, n_early, device=dev),
                        torch.linspace(t_early, t_max, nbp - n_early, device=dev)]).reshape(-1, 1)
        boundary_loss = torch.zeros((), device=dev)
        if self.dimension == 1:
            x_lo, x_hi = self.config.domain[0]
            # both walls in ONE jets launch (streams: u, u_t, u_x); per-point results do not depend on the batch
            xw = torch.cat([torch.full((nbp, 1), x_lo, device=dev), torch.full((nbp, 1), x_hi, device=dev)], dim=0)
            jw = model.jets(xw, torch.cat([tb, tb], dim=0), 1, 1)
            boundary_loss = boundary_loss + self._apply_loss_fn((jw[0, :nbp] - jw[0, nbp:]).unsqueeze(1))
            boundary_loss = boundary_loss + self._apply_loss_fn((jw[2, :nbp] - jw[2, nbp:]).unsqueeze(1))
        else:
            per_axis = max(nbp // (2 * self.dimension), 1)
            for axis in range(self.dimension):
                free = torch.empty(per_axis, self.dimension, device=dev)
                for d in range(self.dimension):
                    lo, hi = self.config.domain[d]
                    free[:, d] = torch.rand(per_axis, device=dev) * (hi - lo) + lo
                ta = torch.rand(per_axis, 1, device=dev) * (t_max - self.config.time_domain[0]) + self.config.time_domain[0]
                cmin, cmax = free.clone(), free.clone()
                cmin[:, axis], cmax[:, axis] = self.config.domain[axis]
                boundary_loss = boundary_loss + self._apply_loss_fn(
                    model(torch.cat([cmin, ta], dim=1)) - model(torch.cat([cmax, ta], dim=1)))
        nip = self._num_points("num_initial_points", 5, len(x))
        if self.dimension == 1:
            x_lo, x_hi = self.config.domain[0]
            xb = (x_hi - x_lo) * 0.1
            xi = torch.cat([torch.linspace(x_lo, x_lo + xb, nip // 4, device=dev),
                            torch.linspace(x_lo + xb, x_hi - xb, nip // 2, device=dev),
                            torch.linspace(x_hi - xb, x_hi, nip // 4, device=dev)]).reshape(-1, 1)
        else:
            xi = torch.empty(nip, self.dimension, device=dev)
            for d in range(self.dimension):
                lo, hi = self.config.domain[d]
                xi[:, d] = torch.rand(nip, device=dev) * (hi - lo) + lo
        ti = torch.zeros(xi.shape[0], 1, device=dev)
        ui = model(torch.cat([xi, ti], dim=1))
        if "initial" in self.boundary_conditions:
            target = self.boundary_conditions["initial"](xi, ti)
        else:
            k = self.config.initial_condition.get("frequency", 2.0)
            target = torch.ones(xi.shape[0], 1, device=dev)
            for d in range(self.dimension):
                target = target * torch.sin(k * torch.pi * xi[:, d : d + 1])
        initial_loss = self._apply_loss_fn(ui - target)
        lw = self._loss_weights()
        smoothness_loss = torch.zeros((), device=dev)
        if lw and lw.get("smoothness", 0.0) > 0:
            smoothness_loss = self._compute_smoothness_loss(model, x, t)
        return self._compose_losses(residual_loss, boundary_loss, initial_loss, smoothness_loss,
                                    self._compute_data_loss(model), aux_scale)

    def _manual_chain(self, n_batch: int):
        """heat_equation.py:375-540 as a launch-list description (see PDEBase._manual_chain), 1-D: ONE (u, u_t, u_x) jet
        launch over [left wall | right wall | initial points]; the periodic boundary loss is two PAIRED terms (value and
        d/dx of the two walls at the same clustered times), the initial loss a target term."""
        if self.dimension != 1:
            raise NotImplementedError("HeatEquation launch list: 1-D only (the >= 2-D boundary points are re-drawn every step)")
        dev = self.device
        nbp = self._num_points("num_boundary_points", 10, n_batch)
        nip = self._num_points("num_initial_points", 5, n_batch)
        t_max = self.config.time_domain[1]
        t_early = t_max * 0.01
        n_early = max(nbp // 4, 1)
        tb = torch.cat([torch.linspace(0, t_early, n_early, device=dev),
                        torch.linspace(t_early, t_max, nbp - n_early, device=dev)]).reshape(-1, 1)
        x_lo, x_hi = self.config.domain[0]
        xb = (x_hi - x_lo) * 0.1
        xi = torch.cat([torch.linspace(x_lo, x_lo + xb, nip // 4, device=dev),
                        torch.linspace(x_lo + xb, x_hi - xb, nip // 2, device=dev),
                        torch.linspace(x_hi - xb, x_hi, nip // 4, device=dev)]).reshape(-1, 1)
        ti = torch.zeros(xi.shape[0], 1, device=dev)
        if "initial" in self.boundary_conditions:
            target = self.boundary_conditions["initial"](xi, ti)
        else:
            target = torch.sin(self.config.initial_condition.get("frequency", 2.0) * torch.pi * xi)
        lw = self._loss_weights()
        bw, iw = (lw.get("boundary", 10.0), lw.get("initial", 10.0)) if lw else (10.0, 10.0)
        x_all = torch.cat([torch.full((nbp, 1), x_lo, device=dev), torch.full((nbp, 1), x_hi, device=dev), xi], 0).contiguous()
        t_all = torch.cat([tb, tb, ti], 0).contiguous()
        terms = [(0, nbp, 0, nbp, None, float(bw)), (0, nbp, 2, nbp, None, float(bw)),  # u and u_x (stream 2 of [u, u_t, u_x])
                 (2 * nbp, 2 * nbp + xi.shape[0], 0, 0, target.reshape(-1).float().contiguous(), float(iw))]
        return {"x": x_all, "t": t_all, "nt": 1, "nx": 1, "terms": terms, "n_bc": 2}

    def _compute_smoothness_loss(self, model, x, t):  # heat_equation.py:625-650
        eps = 1e-4
        x, t = x.detach(), t.detach()
        uc = model(torch.cat([x, t], dim=1))
        out = torch.zeros((), device=self.device)
        for d in range(self.dimension):
            xp, xm = x.clone(), x.clone()
            xp[:, d : d + 1] = torch.clamp(x[:, d : d + 1] + eps, self.domain[d][0], self.domain[d][1])
            xm[:, d : d + 1] = torch.clamp(x[:, d : d + 1] - eps, self.domain[d][0], self.domain[d][1])
            up, um = model(torch.cat([xp, t], dim=1)), model(torch.cat([xm, t], dim=1))
            out = out + torch.mean(torch.abs((up - uc) / eps)) + torch.mean(torch.abs((uc - um) / eps))
        return out

    def exact_solution(self, x, t):  # heat_equation.py:112-196 (1-D / product forms)
        es = getattr(self.config, "exact_solution", None) or {}
        src = es if es else (getattr(self.config, "initial_condition", None) or {})
        A, k = src.get("amplitude", 1.0), src.get("frequency", 2.0)
        dr = self._calculate_decay_rate(k)
        if es.get("type") == "sine_2d" and self.dimension == 2:
            kx, ky = es.get("frequency_x", 2.0), es.get("frequency_y", 2.0)
            tf = torch.exp(-self.alpha * ((kx * torch.pi) ** 2 + (ky * torch.pi) ** 2) * t)
            return A * tf * torch.sin(kx * torch.pi * x[:, 0:1]) * torch.sin(ky * torch.pi * x[:, 1:2])
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            Ld = self.config.domain[d][1] - self.config.domain[d][0]
            sol = sol * torch.sin(2 * torch.pi * k / Ld * x[:, d : d + 1])
        return A * torch.exp(-dr * t) * sol


class AllenCahnEquation(PDEBase):
    """u_t - eps^2 u_xx - u + u^3  (allen_cahn.py:39-111; caller tensors get requires_grad_ in place)."""

    KIND = "allen_cahn"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def epsilon(self):
        return self.get_parameter("epsilon", default=0.1)

    def _coefficients(self):
        return (self.epsilon,)

    def _prepare_model(self, model):  # allen_cahn.py does not touch the model's mode or parameter flags
        pass

    def compute_residual(self, model, x, t):
        if x.is_leaf:
            x.requires_grad_(True)  # allen_cahn.py:51-52 side effect on the caller's tensors
        if t.is_leaf:
            t.requires_grad_(True)
        return super().compute_residual(model, x, t)

    def _residual_from_jets(self, j, x, nt, nx):
        u = j[0]
        if self.dimension > 1:
            return j[1] - u + u**3
        return j[1] - self.epsilon**2 * j[nt + 2] - u + u**3

    def _create_boundary_condition(self, bc_type, params):  # allen_cahn.py:131-151
        if bc_type == "initial":
            kind = params.get("type", "tanh")
            if kind == "tanh":
                if self.dimension == 1:
                    return lambda x, t: torch.tanh(x / (2 * self.epsilon))
                return lambda x, t: torch.tanh(torch.sum(x, dim=1, keepdim=True) / (2 * self.epsilon))
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # allen_cahn.py:113-129
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            sol = sol * torch.tanh(x[:, d : d + 1] / (2 * self.epsilon))
        return sol


class KdVEquation(PDEBase):
    """u_t + 6 u u_x + u_xxx  (kdv_equation.py:38-92; the YAML's alpha/beta are unused upstream)."""

    KIND = "kdv"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def speed(self):
        return self.get_parameter("speed", default=1.0)

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:
            return j[1]
        return j[1] + 6 * j[0] * j[nt + 1] + j[nt + 3]

    def _create_boundary_condition(self, bc_type, params):  # kdv_equation.py:114-141
        if bc_type == "initial":
            kind = params.get("type", "soliton")
            if kind == "soliton":
                c = torch.tensor(params.get("speed", _as_float(self.speed)), dtype=torch.float32, device=self.device)
                if self.dimension == 1:
                    return lambda x, t: 2 * c * (1 / torch.cosh(torch.sqrt(c) * x)) ** 2
                return lambda x, t: 2 * c * (1 / torch.cosh(torch.sqrt(c) * torch.sum(x, dim=1, keepdim=True))) ** 2
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # kdv_equation.py:94-112
        if not self.config.exact_solution:
            return None
        c = torch.tensor(_as_float(self.speed), dtype=x.dtype, device=x.device)
        xs = x if self.dimension == 1 else torch.sum(x, dim=1, keepdim=True)
        return 2 * c * (1 / torch.cosh(torch.sqrt(c) * (xs - c * t))) ** 2


class CahnHilliardEquation(PDEBase):
    """1-D: u_t - d_xx(-eps^2 u_xx + c^3 - c), c = clamp(u, +-10); >= 2-D: u_t  (cahn_hilliard.py:39-160)."""

    KIND = "cahn_hilliard"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def epsilon(self):
        return self.get_parameter("epsilon", default=0.1)

    def _coefficients(self):
        return (self.epsilon,)

    def _prepare_model(self, model):
        pass

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:
            return j[1]
        u = j[0]
        m = ((u >= -10.0) & (u <= 10.0)).to(u.dtype)
        c = torch.clamp(u, -10.0, 10.0)
        ux, uxx = j[nt + 1], j[nt + 2]
        return j[1] + self.epsilon**2 * j[nt + 4] - m * (6 * c * ux**2 + (3 * c**2 - 1) * uxx)

    def _create_boundary_condition(self, bc_type, params):  # cahn_hilliard.py:180-203
        if bc_type == "initial":
            kind = params.get("type", "tanh")
            if kind == "tanh":
                if self.dimension == 1:
                    return lambda x, t: torch.tanh(x / (2 * self.epsilon))
                return lambda x, t: torch.tanh(torch.sum(x, dim=1, keepdim=True) / (2 * self.epsilon))
            if kind == "random":
                amp = params.get("amplitude", 0.1)
                return lambda x, t: amp * (2 * torch.rand_like(x[:, 0:1]) - 1)
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # cahn_hilliard.py:162-178
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            sol = sol * torch.tanh(x[:, d : d + 1] / (2 * self.epsilon))
        return sol


class WaveEquation(PDEBase):
    """u_tt - c^2 u_xx  (wave_equation.py:38-119)."""

    KIND = "wave"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def c(self):
        return self.get_parameter("c", default=1.0)

    def _coefficients(self):
        return (self.c,)

    def _prepare_model(self, model):
        pass

    def compute_residual(self, model, x, t):
        if x.is_leaf:
            x.requires_grad_(True)  # wave_equation.py:50-51
        if t.is_leaf:
            t.requires_grad_(True)
        return super().compute_residual(model, x, t)

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:
            return j[2]
        return j[2] - self.c**2 * j[nt + 2]

    def _create_boundary_condition(self, bc_type, params):  # wave_equation.py:138-168
        if bc_type == "initial":
            kind = params.get("type", "sine")
            if kind == "sine":
                A, k = params.get("amplitude", 1.0), params.get("frequency", 2.0)
                if self.dimension == 1:
                    return lambda x, t: A * torch.sin(k * torch.pi * x)
                return lambda x, t: A * torch.sin(k * torch.pi * torch.sum(x, dim=1, keepdim=True))
            if kind == "sine_2d" and self.dimension == 2:
                A, kx, ky = params.get("amplitude", 1.0), params.get("frequency_x", 2.0), params.get("frequency_y", 2.0)
                return lambda x, t: A * torch.sin(kx * torch.pi * x[:, 0:1]) * torch.sin(ky * torch.pi * x[:, 1:2])
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # wave_equation.py:121-136
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            sol = sol * torch.sin(2 * torch.pi * (x[:, d : d + 1] - self.c * t))
        return sol


class ConvectionEquation(PDEBase):
    """u_t + v u_x  (convection_equation.py:43-78)."""

    KIND = "convection"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def velocity(self):
        v = self.get_parameter("velocity", default=1.0)
        return [v] * self.dimension if isinstance(v, (int, float)) else v

    def _coefficients(self):
        return (self.velocity[0],)

    def _prepare_model(self, model):
        pass

    def _pde_desc(self):
        if self.dimension > 1:
            # convection_equation.py:66-76: for dimension > 1 the reference calls autograd.grad(u, x[:, dim:dim+1]) — a slice that is
            # not part of u's graph — and torch raises exactly this RuntimeError.  Same type, same text, from every residual path.
            raise RuntimeError("One of the differentiated Tensors appears to not have been used in the graph. "
                               "Set allow_unused=True if this is the desired behavior.")
        return super()._pde_desc()

    def _residual_from_jets(self, j, x, nt, nx):
        return j[1] + self.velocity[0] * j[nt + 1]

    def _create_boundary_condition(self, bc_type, params):  # convection_equation.py:97-122
        if bc_type == "initial":
            kind = params.get("type", "sine")
            if kind in ("sine", "sin"):
                A, k = params.get("amplitude", 1.0), params.get("frequency", 2.0)
                if self.dimension == 1:
                    return lambda x, t: A * torch.sin(k * torch.pi * x)
                return lambda x, t: A * torch.sin(k * torch.pi * torch.sum(x, dim=1, keepdim=True))
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # convection_equation.py:80-95
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            sol = sol * torch.sin(2 * torch.pi * (x[:, d : d + 1] - self.velocity[d] * t))
        return sol


class BlackScholesEquation(PDEBase):
    """V_t + 1/2 sigma^2 S^2 V_SS + r S V_S - r V  (black_scholes.py:44-93)."""

    KIND = "black_scholes"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def sigma(self):
        return self.get_parameter("sigma", default=0.2)

    @property
    def r(self):
        return self.get_parameter("r", default=0.05)

    def _coefficients(self):
        return (self.sigma, self.r)

    def _residual_from_jets(self, j, x, nt, nx):
        if self.dimension > 1:  # black_scholes.py:84-91: the spatial terms vanish, -rV does not
            return j[1] - self.r * j[0]
        S = x[:, 0]
        return j[1] + 0.5 * self.sigma**2 * S**2 * j[nt + 2] + self.r * S * j[nt + 1] - self.r * j[0]

    def _create_boundary_condition(self, bc_type, params):  # black_scholes.py:128-150
        if bc_type == "initial":
            kind = params.get("type", "call_option")
            if kind in ("call_option", "option"):
                K = params.get("strike_price", params.get("strike", 1.0))
                if self.dimension == 1:
                    return lambda x, t: torch.maximum(x - K, torch.zeros_like(x))
                return lambda x, t: torch.maximum(torch.sum(x, dim=1, keepdim=True) - K, torch.zeros_like(x[:, 0:1]))
            raise ValueError(f"Unsupported initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # black_scholes.py:95-126 (as written upstream, erf in place of the normal CDF)
        if not self.config.exact_solution:
            return None
        K = self.config.exact_solution.get("strike_price", 1.0)
        sol = torch.ones_like(x[:, 0:1])
        for d in range(self.dimension):
            xd = x[:, d : d + 1]
            d1 = (torch.log(xd / K) + (self.r + 0.5 * self.sigma**2) * t) / (self.sigma * torch.sqrt(t))
            d2 = d1 - self.sigma * torch.sqrt(t)
            sol = sol * (xd * torch.erf(d1) - K * torch.exp(-self.r * t) * torch.erf(d2))
        return sol


class PendulumEquation(PDEBase):
    """u_tt + (g/L) sin u  (pendulum_equation.py:51-94)."""

    KIND = "pendulum"

    def __init__(self, config: PDEConfig, **kwargs):
        super().__init__(config)

    @property
    def g(self):
        return self.get_parameter("g", default=9.81)

    @property
    def L(self):
        return self.get_parameter("L", default=1.0)

    def _coefficients(self):
        g, L = self.g, self.L
        return (g / L,)

    def _residual_from_jets(self, j, x, nt, nx):
        return j[2] + (self.g / self.L) * torch.sin(j[0])

    def _create_boundary_condition(self, bc_type, params):  # pendulum_equation.py:125-155
        if bc_type == "initial":
            kind = params.get("type", "small_angle")
            if kind == "small_angle":
                th = params.get("initial_angle", 0.1)
                return lambda x, t: torch.full_like(x, th)
            if kind == "sine":
                A, f = params.get("amplitude", 1.0), params.get("frequency", 1.0)
                return lambda x, t: A * torch.sin(f * x)
            if kind == "gaussian":
                A, c0, sg = params.get("amplitude", 1.0), params.get("center", 0.0), params.get("sigma", 0.1)
                return lambda x, t: A * torch.exp(-((x - c0) ** 2) / (2 * sg**2))
            raise ValueError(f"Unknown initial condition type: {kind}")
        return super()._create_boundary_condition(bc_type, params)

    def exact_solution(self, x, t):  # pendulum_equation.py:96-123
        if not self.config.exact_solution:
            return None
        kind = self.config.exact_solution.get("type", "small_angle")
        if kind == "small_angle":
            th = self.config.exact_solution.get("initial_angle", 0.1)
            omega = (_as_float(self.g) / _as_float(self.L)) ** 0.5
            return th * torch.cos(omega * t)
        if kind == "sine":
            A, f = self.config.exact_solution.get("amplitude", 1.0), self.config.exact_solution.get("frequency", 1.0)
            return A * torch.sin(f * (x + t))
        raise ValueError(f"Unknown exact solution type: {kind}")

    # ---- the pendulum class's own utilities (pendulum_equation.py:158-212, 232-289): all on the derivative path
    def compute_energy(self, model, x, t):
        """Kinetic + potential energy per point: L^2 (du/dt)^2 / 2 + g L (1 - cos u)."""
        d = self.compute_derivatives(model, x, t, temporal_derivatives=[1], spatial_derivatives=set())
        u = model(torch.cat([x, t], dim=1))
        return 0.5 * self.L * self.L * d["dt"].pow(2) + self.g * self.L * (1 - torch.cos(u))

    def compute_phase_space(self, model, x, t):
        """(theta, d theta / dt)."""
        d = self.compute_derivatives(model, x, t, temporal_derivatives=[1])
        return model(torch.cat([x, t], dim=1)), d["dt"]

    def compute_initial_condition(self, x):
        ic = self.config.initial_condition
        if not ic:
            return None
        kind = ic.get("type", "small_angle")
        if kind == "small_angle":
            return torch.full_like(x, ic.get("initial_angle", 0.1))
        if kind == "sine":
            return ic.get("amplitude", 1.0) * torch.sin(ic.get("frequency", 1.0) * x)
        if kind == "gaussian":
            A, c, sg = ic.get("amplitude", 1.0), ic.get("center", 0.0), ic.get("sigma", 0.1)
            return A * torch.exp(-((x - c) ** 2) / (2 * sg**2))
        raise ValueError(f"Unknown initial condition type: {kind}")

    def compute_boundary_condition(self, x, t):
        bcs = self.config.boundary_conditions
        if not bcs:
            return None
        kind = bcs.get("dirichlet", {}).get("type", "fixed")
        if kind == "fixed":
            return torch.full_like(x, bcs["dirichlet"].get("value", 0.0))
        if kind == "periodic":
            return torch.sin(2 * math.pi * x)
        raise ValueError(f"Unknown boundary condition type: {kind}")
