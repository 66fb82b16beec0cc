"""Import-path mirror of pinnrl/pdes/heat_equation.py."""

from .equations import HeatEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
