"""`PDEConfig` / `PDEBase` with the reference's surface (pinnrl/pdes/pde_base.py), HIP underneath.

What is mirrored: the constructor's normalisation of domain/device/parameters, the
trainable-parameter registry (inverse mode), boundary/initial condition factories,
`compute_derivatives` (key names and the reference's chaining rule), the samplers,
`_apply_loss_fn`, `compute_loss` (term composition and weights) and `validate`.
What changes: derivatives are forward-mode jets from one fused kernel launch instead of chained
`torch.autograd.grad` calls, the residual + loss + weight gradient of `compute_loss` is ONE launch,
and the samplers generate on the device.  Plotting / pickled save-load are out of scope.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import engine as _E

# compiled (time_order, space_order) stream sets, see csrc/Makefile
_STREAM_SETS = [(0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (2, 0), (2, 2)]


@dataclass
class PDEConfig:  # pde_base.py:22-47
    name: str
    domain: Union[Tuple[float, float], List[Tuple[float, float]]]
    time_domain: Tuple[float, float]
    parameters: Dict[str, float]
    boundary_conditions: Dict[str, Dict[str, Any]]
    initial_condition: Dict[str, Any]
    exact_solution: Dict[str, Any]
    dimension: int = 1
    input_dim: Optional[int] = None
    output_dim: Optional[int] = None
    architecture: Optional[str] = None
    device: Optional[torch.device] = None
    training: Optional[Dict[str, Any]] = None
    trainable_parameters: List[str] = field(default_factory=list)
    parameter_initial_guesses: Dict[str, float] = field(default_factory=dict)
    observation_data: Optional[Dict[str, Any]] = None


def _pick_stream_set(nt: int, nx: int) -> Tuple[int, int]:
    best = None
    for a, b in _STREAM_SETS:
        if a >= nt and b >= nx and (best is None or a + b < best[0] + best[1]):
            best = (a, b)
    if best is None:
        raise NotImplementedError(f"pinnrl_amd: no compiled jet kernel covers time order {nt} with space order {nx}")
    return best


def _jets_of(model) -> Any:
    fn = getattr(model, "jets", None)
    if fn is None:
        raise NotImplementedError(
            f"pinnrl_amd: {type(model).__name__} is not a pinnrl_amd network (no fused jet kernel); there is no "
            "eager-autograd fallback"
        )
    return fn


class PDEBase:
    """Base class of the PDE family.  Subclasses set `KIND` and `_coefficients()`."""

    KIND: str = ""  # name in pinnrl_amd._lib.PDE

    @staticmethod
    def create(pde_type: str, config: Optional[PDEConfig] = None, **kwargs) -> "PDEBase":
        """Factory by type name (pde_base.py:55-130): 'heat', 'burgers', 'allen_cahn', ... or '<name>_equation'."""
        from . import _BY_TYPE

        key = pde_type.lower().replace("_equation", "").replace("equation", "").strip("_")
        if key not in _BY_TYPE:
            raise ValueError(f"Could not find PDE implementation for type: {pde_type}")
        if config is None:
            config = PDEConfig(
                name=kwargs.pop("name", _BY_TYPE[key].__name__), domain=kwargs.pop("domain", [(0.0, 1.0)]),
                time_domain=kwargs.pop("time_domain", (0.0, 1.0)), parameters=kwargs.pop("parameters", {}),
                boundary_conditions=kwargs.pop("boundary_conditions", {}),
                initial_condition=kwargs.pop("initial_condition", {}), exact_solution=kwargs.pop("exact_solution", {}),
                dimension=kwargs.pop("dimension", 1), input_dim=kwargs.pop("input_dim", None),
                output_dim=kwargs.pop("output_dim", None), architecture=kwargs.pop("architecture", None),
                device=kwargs.pop("device", None), training=kwargs.pop("training", None))
        return _BY_TYPE[key](config=config, **kwargs)

    # ---------------------------------------------------------------- construction (pde_base.py:132-239)
    def __init__(self, config: PDEConfig, rl_agent=None):
        self.config = config
        self.domain = config.domain
        self.rl_agent = rl_agent
        if isinstance(self.domain, list):
            if len(self.domain) > 0:
                if isinstance(self.domain[0], (list, tuple)):
                    self.domain = [(float(d[0]), float(d[1])) for d in self.domain]
                else:
                    self.domain = [(float(self.domain[0]), float(self.domain[1]))]
        else:
            self.domain = [(0.0, 1.0)]  # pde_base.py:155-157 (a bare tuple falls back to the unit interval)
        self.config.domain = self.domain
        self.time_domain = getattr(config, "time_domain", getattr(config, "t_domain", [0.0, 1.0]))
        if isinstance(self.time_domain, list):
            self.time_domain = tuple(self.time_domain)
        if getattr(config, "device", None) is not None:
            self.device = config.device if isinstance(config.device, torch.device) else torch.device(str(config.device))
        else:
            self.device = torch.device("cpu")
        self.config.device = self.device
        self.dimension = config.dimension
        if getattr(config, "parameters", None) is None:
            config.parameters = {}
        self._trainable_params: nn.ParameterDict = nn.ParameterDict()
        self._true_parameters: Dict[str, float] = {}
        guesses = dict(getattr(config, "parameter_initial_guesses", {}) or {})
        for name in list(getattr(config, "trainable_parameters", []) or []):
            true_val = config.parameters.get(name)
            if true_val is not None:
                self._true_parameters[name] = float(true_val)
            init = guesses.get(name, true_val if true_val is not None else 1.0)
            self._trainable_params[name] = nn.Parameter(torch.tensor(float(init), device=self.device))
        self.observation_data = self._load_observation_data(getattr(config, "observation_data", None))
        self._setup_boundary_conditions()
        self._setup_validation_points()
        self.collocation_history: List[np.ndarray] = []
        if self.config.input_dim is None:
            self.config.input_dim = self.dimension + 1
        if self.config.output_dim is None:
            self.config.output_dim = 1

    def _validate_parameters(self):
        pass

    # ---------------------------------------------------------------- parameters (pde_base.py:246-351)
    def get_parameter(self, name: str, default=None, required: bool = False):
        tr = getattr(self, "_trainable_params", None)
        if tr is not None and name in tr:
            return tr[name]
        if getattr(self.config, "parameters", None) is None:
            if required:
                raise ValueError(f"Required parameter '{name}' not found in config")
            return default
        value = self.config.parameters.get(name, default)
        if value is None and required:
            raise ValueError(f"Required parameter '{name}' not found in config")
        return value

    def trainable_parameters_iter(self):
        return self._trainable_params.parameters() if hasattr(self, "_trainable_params") else iter(())

    def get_trainable_parameter_values(self) -> Dict[str, float]:
        return {n: float(p.detach().cpu().item()) for n, p in getattr(self, "_trainable_params", {}).items()}

    def _training_attr(self, key: str, default):
        tr = getattr(self.config, "training", None)
        if tr is None:
            return default
        return tr.get(key, default) if isinstance(tr, dict) else getattr(tr, key, default)

    def _loss_function_name(self) -> str:
        return self._training_attr("loss_function", "mse")

    def _huber_delta(self) -> float:
        return float(self._training_attr("huber_delta", 1.0))

    def _training_mode(self) -> str:
        return str(self._training_attr("mode", "forward"))

    def _apply_loss_fn(self, error: torch.Tensor) -> torch.Tensor:  # pde_base.py:309-326
        name = self._loss_function_name()
        if name == "mae":
            return torch.mean(torch.abs(error))
        if name == "huber":
            return torch.nn.functional.huber_loss(error, torch.zeros_like(error), reduction="mean", delta=self._huber_delta())
        return torch.mean(error**2)

    def _loss_weights(self):
        """`config.training.loss_weights` when training is an object carrying a non-empty dict (pde_base.py:1180,1214-1218)."""
        tr = getattr(self.config, "training", None)
        if tr is None or isinstance(tr, dict):
            return None
        return getattr(tr, "loss_weights", None) or None

    def _data_loss_weight(self, default: float = 1.0) -> float:
        try:
            lw = self.config.training.loss_weights
            return float(lw.get("data", default)) if isinstance(lw, dict) else float(getattr(lw, "data", default))
        except AttributeError:
            return default

    def _compute_data_loss(self, model) -> torch.Tensor:  # pde_base.py:281-291
        obs = getattr(self, "observation_data", None)
        if not obs:
            return torch.zeros((), device=self.device)
        return self._apply_loss_fn(model(torch.cat([obs["x"], obs["t"]], dim=1)) - obs["u"])

    def _load_observation_data(self, obs_cfg):  # pde_base.py:353-415 (the network-backed "well" source is out of scope)
        if not obs_cfg:
            return None
        if obs_cfg.get("source") == "well":
            raise NotImplementedError("pinnrl_amd: the-Well dataset ingestion is out of scope (needs network access)")
        dev = self.device
        if obs_cfg.get("path"):
            if not os.path.exists(obs_cfg["path"]):
                raise FileNotFoundError(f"Observation data file not found: {obs_cfg['path']}")
            data = np.load(obs_cfg["path"])
            arrs = [np.asarray(data[k], dtype=np.float32) for k in ("x", "t", "u")]
        elif all(k in obs_cfg for k in ("x", "t", "u")):
            raw = [obs_cfg[k] for k in ("x", "t", "u")]
            if all(isinstance(v, torch.Tensor) for v in raw):
                return {k: v.to(dev) for k, v in zip(("x", "t", "u"), raw)}
            arrs = [np.asarray(v, dtype=np.float32) for v in raw]
        else:
            return None
        arrs = [a.reshape(-1, 1) if a.ndim == 1 else a for a in arrs]
        return {k: torch.tensor(a, device=dev) for k, a in zip(("x", "t", "u"), arrs)}

    def generate_synthetic_observations(self, n_points: int = 200, noise_std: float = 0.0, seed: Optional[int] = 0):
        """pde_base.py:417-472 — CPU generator for reproducibility, exact solution at the TRUE parameters."""
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(int(seed))
        cols = []
        for d in range(max(int(self.dimension), 1)):
            lo, hi = self.domain[d]
            cols.append(torch.rand(n_points, 1, generator=gen) * (hi - lo) + lo)
        x = (torch.cat(cols, dim=1) if len(cols) > 1 else cols[0]).to(self.device)
        t0, t1 = self.time_domain[0], self.time_domain[1]
        t = (torch.rand(n_points, 1, generator=gen) * (t1 - t0) + t0).to(self.device)
        saved = self._trainable_params
        try:
            if self._true_parameters:
                self._trainable_params = nn.ParameterDict()
            with torch.no_grad():
                u = self.exact_solution(x, t)
                if noise_std and noise_std > 0:
                    u = u + torch.randn(u.shape, generator=gen).to(self.device) * float(noise_std)
        finally:
            self._trainable_params = saved
        self.observation_data = {"x": x, "t": t, "u": u}
        return self.observation_data

    # ---------------------------------------------------------------- boundary / initial conditions
    def _setup_boundary_conditions(self):  # pde_base.py:474-486 — note the "initial" entry joins the BC dict
        self.boundary_conditions = {}
        if getattr(self.config, "boundary_conditions", None):
            for bc_type, params in self.config.boundary_conditions.items():
                self.boundary_conditions[bc_type] = self._create_boundary_condition(bc_type, params)
        if "initial" not in self.boundary_conditions and hasattr(self.config, "initial_condition"):
            self.boundary_conditions["initial"] = self._create_boundary_condition("initial", self.config.initial_condition)

    def _setup_validation_points(self):
        self.validation_points = None

    def _create_boundary_condition(self, bc_type: str, params: Dict[str, Any]):  # pde_base.py:492-571
        if bc_type in ("left", "right"):
            bc_type = "dirichlet"
        if bc_type in ("dirichlet", "neumann"):
            value = params.get("value", 0.0)
            return lambda x, t: torch.full_like(x[:, 0:1], value)
        if bc_type == "periodic":
            if self.dimension == 1:
                return lambda x, t: torch.sin(2 * torch.pi * x[:, 0:1])
            return lambda x, t: torch.sin(2 * torch.pi * torch.sum(x, dim=1, keepdim=True))
        if bc_type == "initial":
            kind = params.get("type", "sine")
            if kind in ("sine", "sin_exp_decay"):
                amp, freq = params.get("amplitude", 1.0), params.get("frequency", 1.0)
                return lambda x, t: amp * torch.sin(freq * torch.pi * x[:, 0:1])
            if kind == "tanh":
                eps = params.get("epsilon", 0.1)
                return lambda x, t: torch.tanh(x[:, 0:1] / eps)
            if kind == "gaussian":
                mean, std = params.get("mean", 0.0), params.get("std", 0.1)
                return lambda x, t: torch.exp(-((x[:, 0:1] - mean) ** 2) / (2 * std**2))
            if kind == "fixed":
                value = params.get("value", 0.0)
                return lambda x, t: torch.full_like(x[:, 0:1], value)
            if kind == "random":
                amp = params.get("amplitude", 0.1)
                return lambda x, t: amp * (2 * torch.rand_like(x[:, 0:1]) - 1)
            if kind == "small_angle":
                ang = params.get("initial_angle", 0.5)
                return lambda x, t: torch.full_like(x[:, 0:1], ang)
            if kind == "option":
                strike, call = params.get("strike", 100.0), params.get("option_type", "call") == "call"
                if call:
                    return lambda x, t: torch.maximum(x[:, 0:1] - strike, torch.zeros_like(x[:, 0:1]))
                return lambda x, t: torch.maximum(strike - x[:, 0:1], torch.zeros_like(x[:, 0:1]))
            print(f"Warning: Unrecognized initial condition type '{kind}'. Defaulting to zero.")
            return lambda x, t: torch.zeros_like(x[:, 0:1])
        print(f"Warning: Unsupported boundary condition type '{bc_type}'. Defaulting to zero.")
        return lambda x, t: torch.zeros_like(x[:, 0:1])

    # ---------------------------------------------------------------- the hot path
    def _coefficients(self) -> Sequence[Any]:
        """PDE coefficients in the order of PinnPdeDesc.coef; entries may be floats or nn.Parameters."""
        return ()

    def _pde_desc(self):
        coefs = [float(c.detach()) if isinstance(c, torch.Tensor) else float(c) for c in self._coefficients()]
        return _E.pde_desc(self.KIND, self.dimension, coefs, self._loss_function_name(), self._huber_delta())

    def _has_trainable_coefficients(self) -> bool:
        return any(isinstance(c, torch.Tensor) and c.requires_grad for c in self._coefficients())

    def _residual_from_jets(self, jets: torch.Tensor, x: torch.Tensor, nt: int, nx: int) -> torch.Tensor:
        """Per-point residual as torch ops on the (K, N) jets — only used when a PDE coefficient is trainable."""
        raise NotImplementedError

    def _prepare_model(self, model) -> None:
        """Side effects of the reference's residual path: parameters require grad, train mode (pde_base.py:634-638)."""
        if isinstance(model, nn.Module):
            for p in model.parameters():
                p.requires_grad_(True)
            model.train()

    def compute_residual(self, model, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """(N, 1) residual with a grad_fn towards the network parameters — one fused launch.

        Reference: the per-PDE `compute_residual` (see each subclass for file:line).  The coordinates
        are constants of the graph, as after the reference's `detach()`.
        """
        jets_fn = _jets_of(model)
        self._prepare_model(model)
        x = x.detach().to(self.device)
        t = t.detach().to(self.device)
        if self._has_trainable_coefficients():  # inverse mode: jets from the kernel, epilogue in the graph
            nt, nx = _E.pde_streams(self._pde_desc())
            jets = jets_fn(x, t, nt, nx)
            return self._residual_from_jets(jets, x, nt, nx).unsqueeze(1)
        prog = model.program()
        params = prog.tensors
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _E.ResidualFunction.apply(prog, self._pde_desc(), x, t, *params)
        r, _ = _E.residual_forward(prog, self._pde_desc(), x, t)
        return r

    def compute_derivatives(self, model, x: torch.Tensor, t: torch.Tensor, temporal_derivatives: List[int] = None,
                            spatial_derivatives: Set[int] = None) -> Dict[str, torch.Tensor]:
        """pde_base.py:590-794.  Same keys, same chaining rule: the key is the order REQUESTED, the value is
        the derivative obtained after as many chained differentiations as there were requested orders before
        it (so `spatial_derivatives=[2]` alone yields du/dx under "dx2").  In >= 2 spatial dimensions every
        spatial entry is zero, as in the reference (fresh-slice differentiation)."""
        if temporal_derivatives and max(temporal_derivatives) > 2:
            raise ValueError(
                f"Temporal derivative order {max(temporal_derivatives)} is not supported. Maximum order is 2."
            )
        if spatial_derivatives and max(spatial_derivatives) > 4:
            raise ValueError(
                f"Spatial derivative order {max(spatial_derivatives)} is not supported. Maximum order is 4."
            )
        jets_fn = _jets_of(model)
        self._prepare_model(model)
        x = x.detach().to(self.device)
        t = t.detach().to(self.device)
        t_orders = [i for i in sorted(temporal_derivatives or []) if i != 0]
        x_orders = [i for i in sorted(spatial_derivatives or []) if i != 0]
        nt = len(t_orders)
        nx = len(x_orders) if self.dimension == 1 else 0
        try:
            knt, knx = _pick_stream_set(nt, nx)
            jets = jets_fn(x, t, knt, knx)
            t_jets, x_jets = jets, jets
        except NotImplementedError:
            # two time orders together with three or four space orders: no compiled stream set holds both directions.  The
            # streams are pure directional derivatives, so the two directions come from two launches (the reference allows the
            # request: pde_base.py:614-627 only caps the orders)
            a, _ = _pick_stream_set(nt, 0)
            knt, knx = _pick_stream_set(0, nx)
            t_jets = jets_fn(x, t, a, 0)
            x_jets = jets = jets_fn(x, t, knt, knx)
        out: Dict[str, torch.Tensor] = {}
        for c, i in enumerate(t_orders, start=1):
            out["dt" if i == 1 else f"dt{i}"] = t_jets[c].unsqueeze(1)
        if self.dimension == 1:
            for c, i in enumerate(x_orders, start=1):
                out["dx" if i == 1 else f"dx{i}"] = x_jets[knt + c].unsqueeze(1)
            if spatial_derivatives and 2 in spatial_derivatives:
                out["laplacian"] = out["dx2"]
        elif x_orders:
            zero = torch.zeros_like(jets[0]).unsqueeze(1)
            for dim in range(self.dimension):
                name = f"x{dim + 1}"
                for order in x_orders:
                    for i in range(1, order + 1):
                        out[f"d{name * i}"] = zero
            if 2 in x_orders:
                out["laplacian"] = zero
        return out

    def exact_solution(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("Subclasses must implement exact_solution")

    # ---------------------------------------------------------------- samplers (pde_base.py:806-1084), on the device
    def _sample_uniform(self, num_points: int) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = self.device
        if self.dimension == 1:
            n_side = int(np.sqrt(num_points))
            (x0, x1), (t0, t1) = self.domain[0], self.time_domain
            xs = torch.linspace(x0, x1, n_side, device=dev)
            ts = torch.linspace(t0, t1, n_side, device=dev)
            X, T = torch.meshgrid(xs, ts, indexing="ij")
            x = X.reshape(-1, 1)
            t = T.reshape(-1, 1)
            x = x + torch.randn_like(x) * ((x1 - x0) * 0.01)
            t = t + torch.randn_like(t) * ((t1 - t0) * 0.01)
            return torch.clamp(x, x0, x1), torch.clamp(t, t0, t1)
        # >= 2-D: the reference builds the grid, the permutation and the jitter on the CPU generator and moves the
        # result (pde_base.py:829-858); generating on `dev` keeps the step free of host copies.  On a CPU device the
        # op sequence is the reference's, so a shared seed gives its points bit for bit (tests/test_samplers_cpu.py).
        ppd = max(2, int(num_points ** (1 / (self.dimension + 1))) + 1)
        grids = [torch.linspace(self.domain[d][0], self.domain[d][1], ppd, device=dev) for d in range(self.dimension)]
        grids.append(torch.linspace(self.time_domain[0], self.time_domain[1], ppd, device=dev))
        mesh = torch.meshgrid(*grids, indexing="ij")
        pts = torch.stack([g.reshape(-1) for g in mesh], dim=1)
        if len(pts) > num_points:
            pts = pts[torch.randperm(len(pts), device=dev)[:num_points]]
        elif len(pts) < num_points:
            pts = torch.cat([pts, pts[torch.randint(0, len(pts), (num_points - len(pts),), device=dev)]], dim=0)
        pts = pts + torch.randn_like(pts) * 0.01
        for d in range(self.dimension):
            pts[:, d] = torch.clamp(pts[:, d], self.domain[d][0], self.domain[d][1])
        pts[:, -1] = torch.clamp(pts[:, -1], self.time_domain[0], self.time_domain[1])
        return pts[:, : self.dimension].contiguous(), pts[:, -1].reshape(-1, 1).contiguous()

    def _sample_stratified(self, num_points: int) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = self.device
        bounds = [self.domain[d] for d in range(self.dimension)] + [tuple(self.time_domain)]
        samples = torch.zeros(num_points, len(bounds), device=dev)
        for d, (lo, hi) in enumerate(bounds):
            width = (hi - lo) / num_points
            offs = torch.rand(num_points, device=dev)
            idx = torch.arange(num_points, dtype=torch.float32, device=dev)
            col = lo + (idx + offs) * width
            samples[:, d] = col[torch.randperm(num_points, device=dev)]
        return samples[:, : self.dimension].contiguous(), samples[:, -1].reshape(-1, 1).contiguous()

    def _sample_residual_based(self, num_points: int, model=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """RAR, pde_base.py:895-935: 4x pool -> |r| from the forward-only fused kernel -> multinomial."""
        if model is None:
            return self._sample_uniform(num_points)
        x_pool, t_pool = self._sample_uniform(num_points * 4)
        probs = self._residual_sampling_probabilities(model, x_pool, t_pool)
        if probs is None:
            return self._sample_uniform(num_points)
        sel = torch.multinomial(probs, num_points, replacement=True)
        return x_pool[sel].detach(), t_pool[sel].detach()

    def _residual_sampling_probabilities(self, model, x_pool: torch.Tensor, t_pool: torch.Tensor) -> Optional[torch.Tensor]:
        """p_n = (|r_n| + 1e-8) / sum_m (|r_m| + 1e-8) of pde_base.py:917-930.  ONE forward-only launch gives the residual
        field AND its normaliser: with the l1 reduction the fused kernel's loss sum is sum |r| (no extra pass over the
        4N pool).  Trainable coefficients take the jets + torch epilogue path."""
        try:
            with torch.no_grad():
                if self._has_trainable_coefficients() or not hasattr(model, "program"):
                    mag = torch.abs(self.compute_residual(model, x_pool, t_pool).detach()).reshape(-1)
                    total = mag.sum()
                else:
                    _jets_of(model)
                    coefs = [float(c.detach()) if isinstance(c, torch.Tensor) else float(c) for c in self._coefficients()]
                    pd_l1 = _E.pde_desc(self.KIND, self.dimension, coefs, "mae", 1.0)
                    r, total = _E.residual_forward(model.program(), pd_l1, x_pool.detach().to(self.device), t_pool.detach().to(self.device))
                    mag = torch.abs(r).reshape(-1)
        except NotImplementedError:
            raise
        except Exception:
            return None
        return (mag + 1e-8) / (total.reshape(()) + 1e-8 * mag.numel())

    def generate_collocation_points(self, num_points: int, strategy: str = "uniform", **kwargs):
        if strategy == "uniform":
            x, t = self._sample_uniform(num_points)
        elif strategy == "stratified":
            x, t = self._sample_stratified(num_points)
        elif strategy == "residual_based":
            x, t = self._sample_residual_based(num_points, kwargs.get("model", None))
        elif strategy == "adaptive":
            if self.rl_agent is None:
                return self.generate_collocation_points(num_points, strategy="uniform")
            x, t = self._sample_adaptive(num_points)
        else:
            raise ValueError(f"Unknown sampling strategy: {strategy}")
        return x.to(self.device), t.to(self.device)

    def _sample_adaptive(self, num_points: int):
        """pde_base.py:961-1073 — the DQN agent scores a G x G grid; multinomial draw, jitter, clamp."""
        dev = self.device
        G = min(100, max(10, int(np.sqrt(num_points))))
        grids = [torch.linspace(self.domain[d][0], self.domain[d][1], G, device=dev) for d in range(self.dimension)]
        grids.append(torch.linspace(self.time_domain[0], self.time_domain[1], G, device=dev))
        mesh = torch.meshgrid(*grids, indexing="ij")
        points = torch.stack([g.flatten() for g in mesh], dim=1)
        with torch.no_grad():
            probs = torch.abs(self.rl_agent.select_action(points))
            probs = probs / torch.sum(probs)
        idx = torch.multinomial(probs.flatten(), min(num_points, len(points)), replacement=True)
        sel = points[idx]
        if len(sel) < num_points:
            extra = torch.randint(0, len(sel), (num_points - len(sel),), device=dev)
            sel = torch.cat([sel, sel[extra]], dim=0)
        noise_scale = min(0.01, min((self.domain[d][1] - self.domain[d][0]) / G for d in range(self.dimension)),
                          (self.time_domain[1] - self.time_domain[0]) / G)
        sel = sel + torch.randn_like(sel) * noise_scale
        for d in range(self.dimension):
            sel[:, d] = torch.clamp(sel[:, d], self.domain[d][0], self.domain[d][1])
        sel[:, -1] = torch.clamp(sel[:, -1], self.time_domain[0], self.time_domain[1])
        x = sel[:, 0].reshape(-1, 1) if self.dimension == 1 else sel[:, : self.dimension]
        t = sel[:, -1].reshape(-1, 1)
        self.collocation_history.append(sel.cpu().numpy())
        if len(self.collocation_history) > 1:
            self.rl_agent.update_epsilon(len(self.collocation_history))
        return x.contiguous(), t.contiguous()

    def _sample_adaptive_device(self, num_points: int):
        """`_sample_adaptive` without a host round trip (no `.item()`, no history copy), for the autograd-free / captured
        step: the agent's `action_probabilities` (fixed-shape form of select_action + abs / sum, explore branch = all mass
        on grid cell 0, as the reference's one-category multinomial), multinomial, jitter, clamp on the device; epsilon
        decays on the device from the second call on (pde_base.py:1068-1071).  The collocation history of the host
        version (one CPU copy of every batch) is not kept."""
        dev = self.device
        G = min(100, max(10, int(np.sqrt(num_points))))
        cache = getattr(self, "_adaptive_grid", None)
        if cache is None or cache[0] != (G, str(dev)):
            grids = [torch.linspace(self.domain[d][0], self.domain[d][1], G, device=dev) for d in range(self.dimension)]
            grids.append(torch.linspace(self.time_domain[0], self.time_domain[1], G, device=dev))
            mesh = torch.meshgrid(*grids, indexing="ij")
            cache = ((G, str(dev)), torch.stack([g.flatten() for g in mesh], dim=1).contiguous())
            self._adaptive_grid = cache
        points = cache[1]
        probs = self.rl_agent.action_probabilities(points)
        idx = torch.multinomial(probs, min(num_points, len(points)), replacement=True)
        sel = points[idx]
        if len(sel) < num_points:
            extra = torch.randint(0, len(sel), (num_points - len(sel),), device=dev)
            sel = torch.cat([sel, sel[extra]], dim=0)
        noise_scale = min(0.01, min((self.domain[d][1] - self.domain[d][0]) / G for d in range(self.dimension)),
                          (self.time_domain[1] - self.time_domain[0]) / G)
        sel = sel + torch.randn_like(sel) * noise_scale
        cols = [torch.clamp(sel[:, d], self.domain[d][0], self.domain[d][1]) for d in range(self.dimension)]
        tcol = torch.clamp(sel[:, -1], self.time_domain[0], self.time_domain[1])
        self._adaptive_calls = getattr(self, "_adaptive_calls", 0) + 1
        if self._adaptive_calls > 1:
            self.rl_agent.update_epsilon_device()
        x = torch.stack(cols, dim=1).contiguous()
        return x, tcol.reshape(-1, 1).contiguous()

    # ---------------------------------------------------------------- losses (pde_base.py:1086-1235)
    def _residual_loss(self, model, x: torch.Tensor, t: torch.Tensor, n_total: Optional[int] = None) -> torch.Tensor:
        """mean_n l(r_n): residual, reduction AND dL/dtheta in one launch when the coefficients are plain numbers."""
        if self._has_trainable_coefficients():
            coefs = [c if isinstance(c, torch.Tensor) else torch.tensor(float(c), dtype=torch.float32, device=self.device)
                     for c in self._coefficients()]
            if len(coefs) <= 2 and torch.is_grad_enabled() and hasattr(model, "program") and all(c.numel() == 1 for c in coefs):
                # fused: weight gradient AND d loss / d coefficient from the one launch (engine.ResidualLossCoefFunction)
                self._prepare_model(model)
                prog = model.program()
                n = int(n_total) if n_total is not None else x.shape[0]
                kind, dim, lname, delta = self.KIND, self.dimension, self._loss_function_name(), self._huber_delta()
                make_pd = lambda vals: _E.pde_desc(kind, dim, vals, lname, delta)  # noqa: E731
                coefs = [c.to(self.device) for c in coefs]
                return _E.ResidualLossCoefFunction.apply(prog, make_pd, x.detach().to(self.device), t.detach().to(self.device), n,
                                                         len(coefs), *coefs, *prog.tensors)
            loss = self._apply_loss_fn(self.compute_residual(model, x, t))
            if n_total is not None and int(n_total) != x.shape[0]:  # shard of a data-parallel batch: local SUM / global N
                loss = loss * (float(x.shape[0]) / float(n_total))
            return loss
        _jets_of(model)
        self._prepare_model(model)
        prog = model.program()
        x = x.detach().to(self.device)
        t = t.detach().to(self.device)
        n = int(n_total) if n_total is not None else x.shape[0]
        return _E.ResidualLossFunction.apply(prog, self._pde_desc(), x, t, n, *prog.tensors)

    def compute_loss(self, model, x: torch.Tensor, t: torch.Tensor, n_total: Optional[int] = None,
                     aux_scale: float = 1.0) -> Dict[str, torch.Tensor]:
        """pde_base.py:1086-1235.  `n_total` / `aux_scale` are for data-parallel shards (distributed.py): the
        residual mean is taken over the GLOBAL point count and the rank-replicated boundary / initial / data
        terms enter `total` with weight 1/world, so that a SUM all-reduce of the gradients is exact."""
        dev = self.device
        residual_loss = self._residual_loss(model, x, t, n_total)
        inp_b, xb, tb, inp_i, xi, ti = self._boundary_and_initial_points()
        # The reference calls model(inp_b) once per boundary condition and model(inp_i) once (pde_base.py:1133-1141);
        # a point's output does not depend on its launch, so ONE forward (and one backward) over both point sets
        # gives the same numbers with a third of the launches.
        if inp_b.shape[1] == inp_i.shape[1]:
            u_all = model(torch.cat([inp_b, inp_i], dim=0))
            ub_pred, ui = u_all[: inp_b.shape[0]], u_all[inp_b.shape[0]:]
        else:  # dimension > 1: the reference's 1-column boundary points do not fit the model; let it raise as it does there
            ub_pred, ui = model(inp_b), model(inp_i)
        boundary_loss = torch.zeros((), device=dev)
        for bc_func in self.boundary_conditions.values():
            boundary_loss = boundary_loss + self._apply_loss_fn(ub_pred - bc_func(xb, tb))
        if "initial" in self.boundary_conditions:
            target = self.boundary_conditions["initial"](xi, ti)
        else:
            target = self._create_boundary_condition("initial", self.config.initial_condition)(xi, ti)
        initial_loss = self._apply_loss_fn(ui - target)
        data_loss = self._compute_data_loss(model)
        smoothness_loss = torch.zeros((), device=dev)
        return self._compose_losses(residual_loss, boundary_loss, initial_loss, smoothness_loss, data_loss, aux_scale)

    def _boundary_and_initial_points(self):
        """The fixed boundary (2*dim values x 100 times) and initial (100 x, t = 0) evaluation points of
        pde_base.py:1101-1131, 1135-1141.  They depend only on the domain, so they are built once per device and
        reused: no host-to-device copy inside the step (which also keeps the step capturable in a HIP graph)."""
        dev = torch.device(self.device)
        cache = getattr(self, "_bc_ic_points", None)
        if cache is not None and cache[0] == dev:
            return cache[1]
        if self.dimension == 1:
            xb = torch.tensor([self.domain[0][0], self.domain[0][1]], dtype=torch.float32, device=dev).reshape(-1, 1)
        else:
            vals: List[float] = []
            for d in range(self.dimension):
                vals.extend([self.domain[d][0], self.domain[d][1]])
            xb = torch.tensor(vals, dtype=torch.float32, device=dev).reshape(-1, 1)
        tb = torch.linspace(self.time_domain[0], self.time_domain[1], 100, device=dev).reshape(-1, 1)
        xb = xb.repeat_interleave(len(tb), dim=0)
        tb = tb.repeat(len(xb) // len(tb), 1)
        inp_b = torch.cat([xb, tb], dim=1)
        xi = torch.linspace(self.domain[0][0], self.domain[0][1], 100, device=dev).reshape(-1, 1)
        ti = torch.zeros_like(xi)
        inp_i = torch.cat([xi, ti], dim=1)
        self._bc_ic_points = (dev, (inp_b, xb, tb, inp_i, xi, ti))
        return self._bc_ic_points[1]

    def _manual_chain(self, n_batch: int) -> Dict[str, Any]:
        """The boundary / initial part of `compute_loss` (pde_base.py:1101-1165) as data for the autograd-free step
        (`PDETrainer._manual_launches`): fixed points, the stream set to evaluate on them, and loss terms
        (lo, hi, stream, pair_offset, target | None, weight) in the form of `engine.jet_losses`; the first `n_bc` terms are the
        boundary loss, the rest the initial loss."""
        inp_b, xb, tb, inp_i, xi, ti = self._boundary_and_initial_points()
        nb, ni = xb.shape[0], xi.shape[0]
        lw = self._loss_weights()
        bw, iw = (lw.get("boundary", 10.0), lw.get("initial", 10.0)) if lw else (10.0, 10.0)
        terms = [(0, nb, 0, 0, fn(xb, tb).reshape(-1).float().contiguous(), float(bw)) for fn in self.boundary_conditions.values()]
        ic_fn = self.boundary_conditions.get("initial") or self._create_boundary_condition("initial", self.config.initial_condition)
        terms.append((nb, nb + ni, 0, 0, ic_fn(xi, ti).reshape(-1).float().contiguous(), float(iw)))
        return {"x": torch.cat([xb, xi], 0).contiguous(), "t": torch.cat([tb, ti], 0).contiguous(), "nt": 0, "nx": 0,
                "terms": terms, "n_bc": len(terms) - 1}

    def _compose_losses(self, residual_loss, boundary_loss, initial_loss, smoothness_loss, data_loss, aux_scale=1.0):
        """The weighting / mode gating tail shared by every `compute_loss` (pde_base.py:1168-1235, heat_equation.py:543-623)."""
        lw_obj = self._loss_weights()
        smoothness_weight = lw_obj.get("smoothness", 0.0) if lw_obj else 0.0
        losses = {"residual": residual_loss, "boundary": boundary_loss, "initial": initial_loss,
                  "smoothness": smoothness_loss, "data": data_loss}
        data_weight = self._data_loss_weight(1.0)
        mode = self._training_mode()
        active = 0.0 if mode == "data_only" else 1.0
        if mode in ("inverse", "data_only", "data_augmented") and data_weight <= 0.0:
            data_weight = 1.0
        tr = self.config.training
        aw = getattr(tr, "adaptive_weights", None) if tr is not None and not isinstance(tr, dict) else None
        if aw is not None and aw.enabled:
            total = active * residual_loss + aux_scale * (
                active * boundary_loss + active * initial_loss + smoothness_weight * smoothness_loss
                + data_weight * data_loss)
        else:
            if lw_obj:
                rw = lw_obj.get("pde", lw_obj.get("residual", 1.0))
                bw, iw = lw_obj.get("boundary", 10.0), lw_obj.get("initial", 10.0)
            else:
                rw, bw, iw = 1.0, 10.0, 10.0
            total = active * rw * residual_loss + aux_scale * (
                active * bw * boundary_loss + active * iw * initial_loss + smoothness_weight * smoothness_loss
                + data_weight * data_loss)
        losses["total"] = total
        return losses

    def validate(self, model, num_points: int = 1000) -> Dict[str, float]:  # pde_base.py:1286-1303
        x, t = self.generate_collocation_points(num_points)
        with torch.no_grad():
            u_pred = model(torch.cat([x, t], dim=1))
        err = torch.abs(u_pred - self.exact_solution(x, t))
        return {"l2_error": torch.mean(err**2).item(), "max_error": torch.max(err).item(),
                "mean_error": torch.mean(err).item()}

    def update_sampling_strategy(self, x, t, residual):  # pde_base.py:1364-1377 (no caller upstream either)
        reward = torch.mean(torch.abs(residual))
        self.rl_agent.update(torch.cat([x, t], dim=1), reward)
