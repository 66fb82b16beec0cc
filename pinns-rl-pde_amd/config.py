"""Configuration objects the hot path reads — field-compatible with `pinnrl.config`.

Only the boundary is mirrored (pinnrl/config/__init__.py:12-253): the dataclasses whose
attributes `PINNModel`, the PDE classes and `PDETrainer` read, with the same names, defaults
and dict-like `.get()/[]` access, so that the reference's own objects can be passed instead.
The YAML loader / validator (config/__init__.py:363-794) is orchestration and out of scope.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch


@dataclass
class LearningRateSchedulerConfig:  # config/__init__.py:12-28
    type: str = "cosine"
    warmup_epochs: int = 0
    min_lr: float = 1e-6
    factor: float = 0.5
    patience: int = 10


@dataclass
class EarlyStoppingConfig:  # config/__init__.py:31-43
    enabled: bool = True
    patience: int = 10
    min_delta: float = 1e-6


@dataclass
class LBFGSConfig:  # config/__init__.py:46-64
    history_size: int = 50
    max_iter: int = 20
    line_search_fn: Optional[str] = "strong_wolfe"
    tolerance_grad: float = 1e-7
    tolerance_change: float = 1e-9


@dataclass
class AdaptiveWeightsConfig:  # config/__init__.py:67-87
    enabled: bool = False
    strategy: str = "rbw"
    alpha: float = 0.9
    eps: float = 1e-5
    initial_weights: List[float] = None

    def __post_init__(self):
        if self.initial_weights is None:
            self.initial_weights = [0.5, 0.3, 0.2]


@dataclass
class TrainingConfig:  # config/__init__.py:90-169
    num_epochs: int = 100
    batch_size: int = 1024
    num_collocation_points: int = 1024
    num_boundary_points: int = 200
    num_initial_points: int = 100
    learning_rate: float = 1e-3
    weight_decay: float = 0.0
    gradient_clipping: float = 1.0
    early_stopping: EarlyStoppingConfig = None
    learning_rate_scheduler: LearningRateSchedulerConfig = None
    collocation_distribution: str = "uniform"
    adaptive_weights: AdaptiveWeightsConfig = None
    loss_weights: Dict[str, float] = None
    optimizer: str = "adam"
    adam_lbfgs_switch_ratio: float = 0.7
    lbfgs: Optional[LBFGSConfig] = None
    mode: str = "forward"
    loss_function: str = "mse"
    huber_delta: float = 1.0

    def __post_init__(self):
        if self.early_stopping is None:
            self.early_stopping = EarlyStoppingConfig()
        if self.learning_rate_scheduler is None:
            self.learning_rate_scheduler = LearningRateSchedulerConfig()
        if self.loss_weights is None:
            self.loss_weights = {"residual": 1.0, "boundary": 1.0, "initial": 1.0}
        if "data" not in self.loss_weights:
            self.loss_weights["data"] = 1.0
        if self.adaptive_weights is None:
            self.adaptive_weights = AdaptiveWeightsConfig()
        if self.lbfgs is None:
            self.lbfgs = LBFGSConfig()
        if self.optimizer not in ("adam", "lbfgs", "adam_lbfgs"):
            raise ValueError(f"Invalid optimizer '{self.optimizer}'. Choose from 'adam', 'lbfgs', or 'adam_lbfgs'.")
        if self.mode not in ("forward", "inverse", "data_only", "data_augmented"):
            raise ValueError(
                f"Invalid mode '{self.mode}'. Choose 'forward', 'inverse', 'data_only', or 'data_augmented'."
            )
        if self.loss_function not in ("mse", "mae", "huber"):
            raise ValueError(f"Invalid loss_function '{self.loss_function}'. Choose 'mse', 'mae', or 'huber'.")

    @property
    def optimizer_config(self) -> Dict[str, Any]:
        return {"learning_rate": self.learning_rate, "weight_decay": self.weight_decay}

    def __getitem__(self, key: str) -> Any:
        if key == "optimizer_config":
            return self.optimizer_config
        return getattr(self, key)

    def get(self, key: str, default: Any = None) -> Any:
        if key == "optimizer_config":
            return self.optimizer_config
        return getattr(self, key, default)


class ModelConfig:
    """config/__init__.py:172-253 — hand-written __init__, class-level defaults for the optional fields.

    As in the reference, `omega_0`, `num_heads`, ... default to None as CLASS attributes, so
    `config.get("omega_0", 30.0)` returns None unless the field was set explicitly.
    """

    hidden_dims: Optional[List[int]] = None
    omega_0: Optional[float] = None
    num_blocks: Optional[int] = None
    num_heads: Optional[int] = None
    latent_dim: Optional[int] = None
    mapping_size: int = 32
    scale: float = 10.0
    modes: Optional[int] = None

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, activation: str,
                 fourier_features: int = 0, fourier_scale: float = 1.0, dropout: float = 0.0,
                 layer_norm: bool = False, architecture: str = "feedforward"):
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.num_layers = num_layers
        self.activation = activation
        self.fourier_features = fourier_features
        self.fourier_scale = fourier_scale
        self.dropout = dropout
        self.layer_norm = layer_norm
        self.architecture = architecture
        self.hidden_dims = [hidden_dim] * num_layers  # config/__init__.py:241
        if architecture in ("resnet", "fno"):
            self.num_blocks = num_layers  # config/__init__.py:244-245

    def get(self, key: str, default: Any = None) -> Any:
        return getattr(self, key, default)

    def __getitem__(self, key: str) -> Any:
        return getattr(self, key)


class Config:
    """Minimal stand-in for `pinnrl.config.Config`: `.device`, `.model`, `.training` (+ dict-like access).

    Tests and notebooks of the reference build it as `Config.__new__(Config)` and fill the
    attributes by hand (tests/unit_tests/test_pde_arch_matrix.py:40-72); that keeps working.
    """

    def __init__(self, model: Optional[ModelConfig] = None, training: Optional[TrainingConfig] = None,
                 device: Optional[torch.device] = None):
        self.model = model
        self.training = training if training is not None else TrainingConfig()
        self.device = device if device is not None else default_device()
        self.pde = None
        self.rl = None
        self.paths = None

    def get(self, key: str, default: Any = None) -> Any:
        return getattr(self, key, default)

    def __getitem__(self, key: str) -> Any:
        return getattr(self, key)


def default_device() -> torch.device:
    """config/__init__.py:676-690 picks cuda -> mps -> cpu; on ROCm `cuda` IS the MI355X."""
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")
