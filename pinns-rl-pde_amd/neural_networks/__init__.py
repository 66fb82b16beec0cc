"""Network classes with the reference's names, constructor arguments, parameter layout and
`state_dict` keys (pinnrl/neural_networks/*.py) — but whose arithmetic is the fused HIP jet
kernel.  The `nn.Linear` / `nn.LayerNorm` sub-modules are parameter containers only: they fix
the initialisation order (same theta_0 under the same `torch.manual_seed` as the reference
built on CPU) and the checkpoint format; `forward` never calls them.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import engine as _E

InputType = Union[torch.Tensor, np.ndarray, List]
OutputType = Union[torch.Tensor, np.ndarray, List]
NetworkConfig = Dict[str, Any]

__all__ = [
    "BaseNetwork", "InputType", "OutputType", "NetworkConfig", "FeedForwardNetwork", "ResNet", "ResNetBlock",
    "SIREN", "SIRENLayer", "FourierNetwork", "FourierFeatures", "AttentionNetwork", "SelfAttention",
    "FeedForwardBlock", "PINNModel",
]

_ACT_MODULES = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "gelu": nn.GELU}


def _cfg_get(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return config.get(key, default) if hasattr(config, "get") else getattr(config, key, default)


class BaseNetwork(nn.Module):
    """base_network.py:15-104 + the engine hooks (`program`, `jets`)."""

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        dev = _cfg_get(config, "device", torch.device("cpu"))
        self.device = torch.device(dev) if not isinstance(dev, torch.device) else dev
        self._prog_cache = None

    # ---- reference surface -----------------------------------------------------------------
    def _prepare_input(self, x: InputType) -> torch.Tensor:  # base_network.py:41-58
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, dtype=torch.float32, device=self.device)
        if x.device != self.device:
            x = x.to(self.device)
        return x

    def save_state(self, path: str) -> None:  # base_network.py:60-67 (config stored as plain data)
        cfg = self.config if isinstance(self.config, dict) else {
            k: v for k, v in vars(self.config).items() if isinstance(v, (int, float, str, bool, list, type(None)))
        }
        cfg = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list, type(None)))}
        torch.save({"model_state_dict": self.state_dict(), "config": cfg}, path)

    def load_state(self, path: str) -> None:  # base_network.py:69-77, with a loader that executes nothing
        state = torch.load(path, map_location=self.device, weights_only=True)
        self.load_state_dict(state["model_state_dict"])

    def count_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_model_summary(self) -> Dict:
        return {
            "num_parameters": self.count_parameters(),
            "device": str(self.device),
            "memory_usage": f"{sum(p.numel() * p.element_size() for p in self.parameters()) / 1024**2:.2f} MB",
        }

    def _get_activation_module(self, activation_name: str) -> nn.Module:  # base_network.py:91-104
        if activation_name not in _ACT_MODULES:
            raise ValueError(f"Unsupported activation: {activation_name}")
        return _ACT_MODULES[activation_name]()

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        try:
            p = next(self.parameters())
            self.device = p.device
            for m in self.modules():
                if isinstance(m, BaseNetwork) or hasattr(m, "device") and isinstance(getattr(m, "device"), torch.device):
                    m.device = p.device
        except StopIteration:
            pass
        return out

    # ---- engine hooks ----------------------------------------------------------------------
    def _program_spec(self) -> Dict[str, Any]:
        """{arch, activation, input_dim, widths, mapping_size, omega_0} of the fused kernel, or raise."""
        raise NotImplementedError(f"{type(self).__name__} has no fused HIP kernel")

    def program(self) -> _E.NetProgram:
        """The C-ABI view of the LIVE parameters (rebuilt when storage moves; in-place updates are seen)."""
        sd = self.state_dict(keep_vars=True)
        tensors = list(sd.values())
        key = tuple(t.data_ptr() for t in tensors)
        if self._prog_cache is None or self._prog_cache[0] != key:
            spec = self._program_spec()
            trainable = [isinstance(t, nn.Parameter) for t in tensors]
            prog = _E.NetProgram(tensors=tensors, trainable=trainable, **spec)
            prog.names = list(sd.keys())
            prog.set_deterministic(getattr(self, "_deterministic", False))
            prog.set_layer_major(getattr(self, "_layer_major", False))
            self._prog_cache = (key, prog)
        return self._prog_cache[1]

    def set_layer_major(self, on: bool = True) -> None:
        """Engine hint (PINN_FLAG_LAYER_MAJOR), kept across rebuilds of the program like `set_deterministic`."""
        self._layer_major = bool(on)
        if self._prog_cache is not None:
            self._prog_cache[1].set_layer_major(self._layer_major)

    def set_deterministic(self, on: bool = True) -> None:
        """Bit-reproducible weight gradients (PINN_FLAG_DETERMINISTIC: fixed-order reductions in the layer-major
        engine); kept across rebuilds of the program.  Reference anchor: tests/unit_tests/test_benchmarks.py:61-64."""
        self._deterministic = bool(on)
        if self._prog_cache is not None:
            self._prog_cache[1].set_deterministic(self._deterministic)

    def jets(self, x: torch.Tensor, t: torch.Tensor, time_order: int = 0, space_order: int = 0) -> torch.Tensor:
        """(K, N) = [u, d/dt.., d/dx..] in one launch; differentiable w.r.t. the parameters."""
        prog = self.program()
        params = prog.tensors
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _E.JetFunction.apply(prog, x, t, time_order, space_order, *params)
        return _E.jets_forward(prog, x, t, time_order, space_order)

    def forward(self, x: InputType) -> OutputType:
        inp = self._prepare_input(x)
        if inp.dim() != 2:
            raise ValueError(f"expected a (N, input_dim) input, got {tuple(inp.shape)}")
        u = self.jets(inp[:, :-1], inp[:, -1:], 0, 0)
        return u[0].unsqueeze(1)


def _dropout_guard(p: float, who: str) -> None:
    if p and p > 0.0:
        raise NotImplementedError(
            f"{who}: dropout={p} makes the residual stochastic (pde_base.py:638 forces train()); the fused HIP path "
            "covers dropout=0.0 only"
        )


# --------------------------------------------------------------------------------------------
class FourierFeatures(nn.Module):  # fourier.py:19-62
    def __init__(self, input_dim: int, mapping_size: int, scale: float = 10.0, device: Optional[torch.device] = None):
        super().__init__()
        self.input_dim, self.mapping_size, self.scale = input_dim, mapping_size, scale
        self.device = device or torch.device("cpu")
        # drawn on the CPU generator (then moved) so theta_0 matches the CPU-built reference for a given seed
        self.register_buffer("B", (torch.randn(input_dim, mapping_size) * scale).to(self.device))
        self.output_dim = mapping_size * 2


class FourierNetwork(BaseNetwork):  # fourier.py:65-124
    def __init__(self, config: NetworkConfig) -> None:
        super().__init__(config)
        self.input_dim = config["input_dim"]
        self.mapping_size = _cfg_get(config, "mapping_size", 32)
        self.hidden_dim = config["hidden_dim"]
        self.num_layers = _cfg_get(config, "num_layers", 4)
        self.output_dim = config["output_dim"]
        self.activation_name = _cfg_get(config, "activation", "relu")
        self.activation_fn = self._get_activation_module(self.activation_name)
        self.scale = _cfg_get(config, "scale", 10.0)
        self.fourier = FourierFeatures(self.input_dim, self.mapping_size, self.scale, device=self.device)
        self.layers = nn.ModuleList()
        prev = 2 * self.mapping_size
        for _ in range(self.num_layers - 1):
            self.layers.append(nn.Linear(prev, self.hidden_dim))
            prev = self.hidden_dim
        self.layers.append(nn.Linear(prev, self.output_dim))
        self.to(self.device)

    def _program_spec(self):
        return dict(arch="fourier", activation=self.activation_name, input_dim=self.input_dim,
                    widths=[self.hidden_dim] * (self.num_layers - 1) + [self.output_dim],
                    mapping_size=self.mapping_size)


class FeedForwardNetwork(BaseNetwork):  # feedforward.py:9-73
    def __init__(self, config: NetworkConfig) -> None:
        super().__init__(config)
        self.input_dim = config["input_dim"]
        self.hidden_dims = list(config["hidden_dims"])
        self.output_dim = config["output_dim"]
        self.dropout_rate = _cfg_get(config, "dropout", 0.1)
        self.use_layer_norm = _cfg_get(config, "layer_norm", True)
        self.activation_name = _cfg_get(config, "activation", "relu")
        layers: List[nn.Module] = []
        prev = self.input_dim
        for h in self.hidden_dims:
            layers.append(nn.Linear(prev, h))
            if self.use_layer_norm:
                layers.append(nn.LayerNorm(h))
            layers.append(self._get_activation_module(self.activation_name))
            if self.dropout_rate > 0.0:
                layers.append(nn.Dropout(self.dropout_rate))
            prev = h
        layers.append(nn.Linear(prev, self.output_dim))
        self.layers = nn.Sequential(*layers)
        self.to(self.device)

    def _program_spec(self):
        _dropout_guard(self.dropout_rate, "FeedForwardNetwork")
        return dict(arch="feedforward", activation=self.activation_name, input_dim=self.input_dim,
                    widths=self.hidden_dims + [self.output_dim], layer_norm=bool(self.use_layer_norm))


class SIRENLayer(nn.Module):  # siren.py:11-46
    def __init__(self, in_features: int, out_features: int, omega_0: float = 30.0) -> None:
        super().__init__()
        self.omega_0 = omega_0
        self.linear = nn.Linear(in_features, out_features)
        with torch.no_grad():
            bound = np.sqrt(6 / in_features) / omega_0
            self.linear.weight.uniform_(-bound, bound)


class SIREN(BaseNetwork):  # siren.py:49-90
    def __init__(self, config: NetworkConfig) -> None:
        super().__init__(config)
        self.input_dim = config["input_dim"]
        self.hidden_dims = list(config["hidden_dims"])
        self.output_dim = config["output_dim"]
        self.omega_0 = _cfg_get(config, "omega_0", 30.0)
        if self.omega_0 is None:  # ModelConfig leaves the class attribute None: same TypeError as upstream
            raise TypeError("SIREN: omega_0 is None — set config.model.omega_0 explicitly")
        self.layers = nn.ModuleList()
        prev = self.input_dim
        for h in self.hidden_dims:
            self.layers.append(SIRENLayer(prev, h, omega_0=self.omega_0))
            prev = h
        self.layers.append(nn.Linear(prev, self.output_dim))
        self.to(self.device)

    def _program_spec(self):
        return dict(arch="siren", activation="sin", input_dim=self.input_dim,
                    widths=self.hidden_dims + [self.output_dim], omega_0=float(self.omega_0))


class ResNetBlock(nn.Module):  # resnet.py:9-65
    def __init__(self, in_dim: int, hidden_dim: int, activation: str = "relu", dropout: float = 0.1):
        super().__init__()
        if activation not in _ACT_MODULES:
            raise ValueError(f"Unsupported activation: {activation}")
        self.activation_fn = _ACT_MODULES[activation]()
        self.layers = nn.Sequential(
            nn.Linear(in_dim, hidden_dim), nn.LayerNorm(hidden_dim), self.activation_fn, nn.Dropout(dropout),
            nn.Linear(hidden_dim, in_dim), nn.LayerNorm(in_dim), nn.Dropout(dropout),
        )


class ResNet(BaseNetwork):  # resnet.py:68-142
    def __init__(self, config: NetworkConfig) -> None:
        super().__init__(config)
        self.input_dim = config["input_dim"]
        if "hidden_dim" in config:
            self.hidden_dim = config["hidden_dim"]
        elif isinstance(config.get("hidden_dims"), list) and config["hidden_dims"]:
            self.hidden_dim = config["hidden_dims"][0]
        else:
            self.hidden_dim = 124
        self.num_blocks = config.get("num_blocks", config.get("num_layers", 4))
        self.output_dim = config["output_dim"]
        self.activation_name = config.get("activation", "relu")
        self.activation_fn = self._get_activation_module(self.activation_name)
        self.dropout = config.get("dropout", 0.1)
        self.input_layer = nn.Linear(self.input_dim, self.hidden_dim)
        self.blocks = nn.ModuleList(
            [ResNetBlock(self.hidden_dim, self.hidden_dim, self.activation_name, self.dropout) for _ in range(self.num_blocks)]
        )
        self.output_layer = nn.Linear(self.hidden_dim, self.output_dim)
        self.to(self.device)

    def _program_spec(self):
        _dropout_guard(self.dropout, "ResNet")
        return dict(arch="resnet", activation=self.activation_name, input_dim=self.input_dim,
                    widths=[self.hidden_dim] * (1 + 2 * self.num_blocks) + [self.output_dim],
                    num_blocks=self.num_blocks)


class SelfAttention(nn.Module):  # attention.py:11-72 — with a length-1 sequence out = LN(proj(value(x)) + x)
    def __init__(self, dim: int, heads: int = 4, dropout: float = 0.1) -> None:
        super().__init__()
        self.dim, self.heads, self.head_dim = dim, heads, dim // heads
        assert self.head_dim * heads == dim, "Dimension must be divisible by heads"
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)
        self.dropout = nn.Dropout(dropout)
        self.scale = self.head_dim**-0.5
        self.layer_norm = nn.LayerNorm(dim)


class FeedForwardBlock(nn.Module):  # attention.py:75-107
    def __init__(self, dim: int, expansion: int = 4, dropout: float = 0.1) -> None:
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, dim * expansion), nn.GELU(), nn.Dropout(dropout), nn.Linear(dim * expansion, dim),
            nn.Dropout(dropout),
        )
        self.layer_norm = nn.LayerNorm(dim)


class AttentionNetwork(BaseNetwork):  # attention.py:110-183
    def __init__(self, config: NetworkConfig) -> None:
        super().__init__(config)
        self.input_dim = config["input_dim"]
        self.hidden_dim = config["hidden_dim"]
        self.output_dim = config["output_dim"]
        self.num_layers = _cfg_get(config, "num_layers", 4)
        self.num_heads = _cfg_get(config, "num_heads", 4) or 4
        self.dropout = _cfg_get(config, "dropout", 0.1)
        self.activation_name = _cfg_get(config, "activation", "gelu")
        self.activation_fn = self._get_activation_module(self.activation_name)
        self.input_proj = nn.Linear(self.input_dim, self.hidden_dim)
        self.layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.layers.append(nn.ModuleList([
                SelfAttention(self.hidden_dim, self.num_heads, self.dropout),
                FeedForwardBlock(self.hidden_dim, dropout=self.dropout),
            ]))
        self.output_proj = nn.Linear(self.hidden_dim, self.output_dim)
        self.apply(self._init_weights)
        self.to(self.device)

    @staticmethod
    def _init_weights(module):  # attention.py:158-163
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()

    def _program_spec(self):
        _dropout_guard(self.dropout, "AttentionNetwork")
        return dict(arch="attention", activation=self.activation_name, input_dim=self.input_dim,
                    widths=[self.hidden_dim, self.output_dim], num_blocks=self.num_layers)


class PINNModel(BaseNetwork):
    """neural_networks/__init__.py:61-154 — architecture factory; `.model` holds the network."""

    def __init__(self, config, device=None, **kwargs):
        self.config = config
        dev = device if device is not None else config.device
        dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
        mc = config.model
        mc.device = dev
        super().__init__(mc)
        self.config = config
        self.device = dev
        self.architecture = mc.architecture
        self.architecture_name = mc.architecture
        a = self.architecture
        if a == "fourier":
            self.model = FourierNetwork(mc)
        elif a == "resnet":
            rc = {"input_dim": mc.input_dim, "hidden_dim": mc.hidden_dim, "output_dim": mc.output_dim,
                  "activation": mc.activation, "dropout": mc.dropout, "device": dev}
            nb = getattr(mc, "num_blocks", None)
            rc["num_blocks"] = nb if nb is not None else mc.num_layers
            if getattr(mc, "hidden_dims", None) is not None:
                rc["hidden_dims"] = mc.hidden_dims
            self.model = ResNet(rc)
        elif a == "siren":
            self.model = SIREN(mc)
        elif a == "attention":
            self.model = AttentionNetwork(mc)
        elif a in ("autoencoder", "fno"):
            raise NotImplementedError(
                f"pinnrl_amd: architecture '{a}' is outside the accelerated hot path (SURVEY.md §2 rows 8-9); "
                "use pinnrl itself for it"
            )
        else:
            self.model = FeedForwardNetwork(mc)
        self.model = self.model.to(dev)
        self.to(dev)

    def _program_spec(self):
        return self.model._program_spec()

    def set_deterministic(self, on: bool = True) -> None:
        self.model.set_deterministic(on)

    def program(self):
        prog = self.model.program()
        return prog

    def jets(self, x, t, time_order: int = 0, space_order: int = 0):
        return self.model.jets(x, t, time_order, space_order)

    def forward(self, x):
        return self.model(x)
