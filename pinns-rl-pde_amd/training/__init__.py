"""Training loop counterpart of `pinnrl.training` (the caller of the hot path)."""

from .trainer import PDETrainer  # noqa: F401
