"""Import shim: exposes the package directory `pinns-rl-pde_amd/` (not a valid Python identifier)
as the importable package `pinnrl_amd`, whose sub-packages mirror `pinnrl`'s
(`pinnrl_amd.neural_networks`, `pinnrl_amd.pdes`, `pinnrl_amd.training`, `pinnrl_amd.config`)."""

import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pinns-rl-pde_amd")
_spec = importlib.util.spec_from_file_location(
    "pinnrl_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pinnrl_amd"] = _mod
_spec.loader.exec_module(_mod)
