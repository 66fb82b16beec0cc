"""Import-path mirror of pinnrl/pdes/convection_equation.py."""

from .equations import ConvectionEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
