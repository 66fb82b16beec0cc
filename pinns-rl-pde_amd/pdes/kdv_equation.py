"""Import-path mirror of pinnrl/pdes/kdv_equation.py."""

from .equations import KdVEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
