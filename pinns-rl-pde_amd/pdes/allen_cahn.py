"""Import-path mirror of pinnrl/pdes/allen_cahn.py."""

from .equations import AllenCahnEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
