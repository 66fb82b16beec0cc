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


# Every relative error a GPU test measures is appended to gpurun_out/parity_r03.jsonl (one JSON object per line: test id,
# label, value, engine-policy environment), so that the margins to the tolerances are on record, not just pass / fail;
# tools/parity_summary.py condenses the file into profiles/parity_r03.md.
_PARITY_LOG = os.environ.get("PINN_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_r03.jsonl"))
_parity_seq = {}


def _record_parity(kind, value, label, tol):
    test = os.environ.get("PYTEST_CURRENT_TEST", "").rsplit(" ", 1)[0]  # "file::test[param] (call)" -> id (ids may contain blanks)
    if not test or not torch.cuda.is_available():
        return
    _parity_seq[test] = _parity_seq.get(test, 0) + 1
    rec = {"test": test, "i": _parity_seq[test], "kind": kind, "label": label, "value": value, "tol": tol,
           "policy": {k: os.environ[k] for k in ("PINN_LM_FUSED", "PINN_LM_FUSED_LN") if k in os.environ}}
    try:
        os.makedirs(os.path.dirname(_PARITY_LOG), exist_ok=True)
        with open(_PARITY_LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def rel_l2(a, b, label=None, tol=None):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    v = float((a - b).norm() / max(float(b.norm()), 1e-30))
    _record_parity("rel_l2", v, label, tol)
    return v


def rel_err(a, b, label=None, tol=None):
    """|a - b| / |b| of two scalars, recorded like rel_l2."""
    a, b = float(a), float(b)
    v = abs(a - b) / max(abs(b), 1e-30)
    _record_parity("rel_err", v, label, tol)
    return v


@pytest.fixture(scope="session")
def golden_cases():
    return CASES
