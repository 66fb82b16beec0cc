"""Import-path mirror of pinnrl/pdes/wave_equation.py."""

from .equations import WaveEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
