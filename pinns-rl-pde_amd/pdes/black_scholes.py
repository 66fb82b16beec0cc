"""Import-path mirror of pinnrl/pdes/black_scholes.py."""

from .equations import BlackScholesEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
