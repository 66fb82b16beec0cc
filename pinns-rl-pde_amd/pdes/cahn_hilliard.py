"""Import-path mirror of pinnrl/pdes/cahn_hilliard.py."""

from .equations import CahnHilliardEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
