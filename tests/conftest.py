"""Shared pytest configuration: the `gpu` marker and golden-fixture helpers."""

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


MANIFEST = _manifest()
CASES = sorted(k for k in MANIFEST if not k.startswith("_"))


def load_case(tag):
    """Golden fixture -> (ArchSpec, PdeSpec, state_dict, arrays, manifest entry)."""
    from oracle import ArchSpec, PdeSpec

    m = MANIFEST[tag]
    z = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    spec = ArchSpec(**m["arch"])
    p = m["pde"]
    pde = PdeSpec(
        name=p["name"], dimension=p["dimension"], domain=[tuple(d) for d in p["domain"]],
        time_domain=tuple(p["time_domain"]), parameters=p["parameters"],
        boundary_conditions=p["boundary_conditions"], initial_condition=p["initial_condition"],
    )
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    arrays = {k: z[k] for k in z.files if not k.startswith("sd::")}
    return spec, pde, sd, arrays, m


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.fixture(scope="session")
def golden_cases():
    return CASES
