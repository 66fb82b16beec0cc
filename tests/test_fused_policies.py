"""The layer-major engine's node kernels come in two implementations: GEMM + element-wise launches (lm_gemm.h / lm_ew.h)
and the fused GEMM + prologue kernels (lm_fused.h); which nodes take which is a measured policy read ONCE per process
(PINN_LM_FUSED, PINN_LM_FUSED_LN, lm_engine.hip::plan_fusion).  The default policy runs in every other GPU test; here
the full-depth / full-width parity tests run again in child processes with everything unfused and with every
LayerNorm node fused (kernels the default policy leaves to the unfused path must stay correct: they are selectable),
and with the two GEMM-side choices of lm_engine.hip switched back (PINN_LM_WRES16, PINN_LM_NT_BATCH)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"PINN_LM_FUSED": "0"}, {"PINN_LM_FUSED_LN": "15"}, {"PINN_LM_FUSED_LN": "0"},
                                 {"PINN_LM_WRES16": "0", "PINN_LM_NT_BATCH": "0"}],
                         ids=["unfused", "all-layernorm-nodes-fused", "no-layernorm-node-fused",
                              "depth-512-gemm-streamed-and-one-dw-launch-per-layer"])
def test_width256_parity_under_policy(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_width256_parity.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


_CHUNK_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle as O
from conftest import rel_l2
from hip_helpers import pde_desc_from_spec, program_from_spec
from pinnrl_amd import engine as E
dev = torch.device("cuda:0")
cases = [("resnet", "allen_cahn", dict(hidden_dim=64, num_layers=2, num_blocks=2, activation="tanh")),
         ("attention", "burgers", dict(hidden_dim=64, num_layers=1, num_heads=4, activation="gelu")),
         ("siren", "kdv", dict(hidden_dim=128, num_layers=3, omega_0=6.0))]
N = 64 * 32 * 2 + 64 * 32 // 2 + 7   # three chunks of 64 tiles (the record target clamps there), ragged last tile
for arch, pde_name, kw in cases:
    spec = O.ArchSpec(architecture=arch, **kw)
    pde = O.PdeSpec(name=pde_name, domain=((-3.0, 3.0),) if pde_name == "kdv" else ((-1.0, 1.0),),
                    parameters={"burgers": {"nu": 0.02}, "kdv": {}, "allen_cahn": {"epsilon": 0.05}}[pde_name])
    sd = O.init_state_dict(spec, seed=71)
    torch.manual_seed(72)
    x = (torch.rand(N, 1) * 2 - 1) * (3.0 if pde_name == "kdv" else 1.0)
    t = torch.rand(N, 1)
    for det in (False, True):
        prog, names = program_from_spec(spec, sd, dev)
        prog.set_layer_major(True)
        prog.set_deterministic(det)
        pd = pde_desc_from_spec(pde)
        flat = E.new_flat_grad(prog, dev)
        r, s = E.residual_loss_grad(prog, pd, x.to(dev), t.to(dev), 1.0 / N, flat, want_residual=True)
        # chunk-free reference: the same engine on three separate launches (<= 64 tiles each), summed
        parts, fsum, ssum = [], E.new_flat_grad(prog, dev), 0.0
        for lo in range(0, N, 2048):
            hi = min(N, lo + 2048)
            rp, sp = E.residual_loss_grad(prog, pd, x[lo:hi].to(dev), t[lo:hi].to(dev), 1.0 / N, fsum, want_residual=True)
            parts.append(rp); ssum += float(sp)
        assert torch.equal(torch.cat(parts), r), (arch, det, "per-point residuals differ between one multi-chunk launch and per-chunk launches")
        assert abs(float(s) - ssum) <= 2e-5 * abs(ssum), (arch, det)
        assert rel_l2(flat.cpu(), fsum.cpu()) <= 2e-5, (arch, det, rel_l2(flat.cpu(), fsum.cpu()))
    idx = torch.linspace(0, N - 1, 200).long()
    sd64 = {k: v.double() for k, v in sd.items()}
    r_o = O.compute_residual(pde, lambda inp: O.network_forward(spec, sd64, inp, "composite"), x[idx].double(), t[idx].double()).detach()
    assert rel_l2(r[idx.to(dev)].cpu(), r_o) <= 1e-5, (arch, rel_l2(r[idx.to(dev)].cpu(), r_o))
print("multi-chunk ok")
"""


@pytest.mark.gpu
def test_multi_chunk_loop_of_the_layer_major_engine():
    """the chunk loop of lm_run (chunk-local record indexing against global point offsets, gradient accumulation
    across chunks before the single unpack, per-chunk deterministic partials) never runs at test sizes with the 1 GB record
    target.  A child process with PINN_LM_RECORD_MB=1 clamps a chunk to 64 tiles: 5 127 points = 3 chunks with a ragged last
    tile; LayerNorm (ResNet), attention (merged W_p W_v gradient) and SIREN nets, default and deterministic mode, against
    per-chunk launches and the fp64 oracle."""
    e = dict(os.environ, PINN_LM_RECORD_MB="1")
    r = subprocess.run([sys.executable, "-c", _CHUNK_SCRIPT], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "multi-chunk ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
