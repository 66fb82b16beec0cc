"""The layer-major engine's node kernels come in two implementations: GEMM + element-wise launches (lm_gemm.h / lm_ew.h)
and the fused GEMM + prologue kernels (lm_fused.h); which nodes take which is a measured policy read ONCE per process
(PINN_LM_FUSED, PINN_LM_FUSED_LN, lm_engine.hip::plan_fusion).  The default policy runs in every other GPU test; here
the full-depth / full-width parity tests run again in child processes with everything unfused and with every
LayerNorm node fused (kernels the default policy leaves to the unfused path must stay correct: they are selectable)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"PINN_LM_FUSED": "0"}, {"PINN_LM_FUSED_LN": "15"}, {"PINN_LM_FUSED_LN": "0"}],
                         ids=["unfused", "all-layernorm-nodes-fused", "no-layernorm-node-fused"])
def test_width256_parity_under_policy(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_width256_parity.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
