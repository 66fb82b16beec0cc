"""Import-path mirror of pinnrl/pdes/burgers_equation.py."""

from .equations import BurgersEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
