"""Partial differential equations — same names and import paths as `pinnrl.pdes`."""

from .equations import (  # noqa: F401
    AllenCahnEquation, BlackScholesEquation, BurgersEquation, CahnHilliardEquation, ConvectionEquation, HeatEquation,
    KdVEquation, PendulumEquation, WaveEquation,
)
from .pde_base import PDEBase, PDEConfig  # noqa: F401

_BY_TYPE = {
    "heat": HeatEquation, "wave": WaveEquation, "burgers": BurgersEquation, "kdv": KdVEquation,
    "cahn_hilliard": CahnHilliardEquation, "allen_cahn": AllenCahnEquation, "black_scholes": BlackScholesEquation,
    "convection": ConvectionEquation, "pendulum": PendulumEquation,
}


def create_pde(config):
    """pinnrl/pdes/__init__.py:17-49 — type -> class switch, defaulting to heat."""
    pde_type = getattr(config, "type", "heat").lower() if hasattr(config, "type") else "heat"
    if pde_type not in _BY_TYPE:
        raise ValueError(f"PDE type not supported: {pde_type}")
    return _BY_TYPE[pde_type](config)
