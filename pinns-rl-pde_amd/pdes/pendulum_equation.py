"""Import-path mirror of pinnrl/pdes/pendulum_equation.py."""

from .equations import PendulumEquation  # noqa: F401
from .pde_base import PDEBase, PDEConfig  # noqa: F401
