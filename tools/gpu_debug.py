"""Diagnostic sweep on the GPU box: per-stream / per-tensor errors of the HIP path vs the golden vectors."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import CASES, load_case, rel_l2
import jet_model as J
from hip_helpers import program_from_spec, pde_desc_from_spec
from pinnrl_amd import engine as E

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
only = sys.argv[1:] 
for tag in CASES:
    spec, pde, sd, a, m = load_case(tag)
    if spec.architecture not in ("fourier", "feedforward", "siren", "resnet", "attention"):
        continue
    if only and tag not in only:
        continue
    try:
        prog, names = program_from_spec(spec, sd, dev)
        pd = pde_desc_from_spec(pde)
        x, t = torch.from_numpy(a["x"]).to(dev), torch.from_numpy(a["t"]).to(dev)
        N = x.shape[0]
        NT, NX = J.pde_streams(pde.name, pde.dimension)
        jets = E.jets_forward(prog, x, t, NT, NX).cpu()
        torch.cuda.synchronize()
        msg = [f"u {rel_l2(jets[0], a['u64']):.1e}"]
        for k, s in {"jet_dt": 1, "jet_dt2": 2, "jet_dx": NT + 1, "jet_dx2": NT + 2, "jet_dx3": NT + 3, "jet_dx4": NT + 4}.items():
            if k in a and s < jets.shape[0] and not (k == "jet_dt2" and NT < 2):
                msg.append(f"{k[4:]} {rel_l2(jets[s], a[k]):.1e}")
        r, s = E.residual_forward(prog, pd, x, t)
        torch.cuda.synchronize()
        msg.append(f"| r {rel_l2(r.cpu(), a['residual64']):.1e} L {abs(float(s)/N-float(a['loss64']))/abs(float(a['loss64'])):.1e}")
        flat = E.new_flat_grad(prog, dev)
        r2, s2 = E.residual_loss_grad(prog, pd, x, t, 1.0 / N, flat, want_residual=True)
        torch.cuda.synchronize()
        grads = E.split_flat_grad(prog, flat)
        by = {n: g for n, g in zip(names, grads) if g is not None}
        got = torch.cat([by[k].flatten().cpu() for k in m["param_names"]])
        msg.append(f"| bwd r {rel_l2(r2.cpu(), a['residual64']):.1e} grad {rel_l2(got, a['grad64']):.1e}")
        if spec.architecture == "resnet":  # exact gradient (composite LayerNorm, fp64): torch's fused LN is inexact at 3rd order
            sd64 = {k: v.double() for k, v in sd.items()}
            x64, t64 = torch.from_numpy(a["x"]).double(), torch.from_numpy(a["t"]).double()
            jj, tp = J.resnet_jets_forward(spec, sd64, torch.cat([x64, t64], 1), NT, NX)
            rr, dr = J.pde_residual(pde.name, pde.parameters, jj, x64[:, 0:1], NT, NX, pde.dimension)
            ge = J.resnet_jets_backward(spec, sd64, tp, [2.0 * rr / N * d for d in dr], NT, NX)
            exact = torch.cat([ge[k].flatten() for k in m["param_names"]])
            msg.append(f"grad-vs-exact {rel_l2(got, exact):.1e}")
        print(f"{tag:40s} " + " ".join(msg), flush=True)
        # per tensor
        off = 0
        det = []
        for k in m["param_names"]:
            n = by[k].numel()
            ref = torch.from_numpy(a["grad64"][off:off + n]); off += n
            det.append(f"{k.replace('model.','')}:{rel_l2(by[k].flatten().cpu(), ref):.0e}")
        print("      " + " ".join(det), flush=True)
    except Exception:
        print(f"{tag:40s} EXCEPTION", flush=True)
        traceback.print_exc()
