import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import oracle as O
from hip_helpers import pde_desc_from_spec, program_from_spec
from pinnrl_amd import engine as E
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
spec = O.ArchSpec(architecture="fourier", input_dim=2, hidden_dim=1, num_layers=4, mapping_size=8, scale=1.0, activation="tanh")
pde = O.PdeSpec(name="kdv", dimension=1, domain=((-3.0, 3.0),), time_domain=(0.0, 1.0), parameters={})
sd = O.init_state_dict(spec, seed=3)
torch.manual_seed(3)
x, t = O.sample_uniform(pde, 10); x, t = x[:5].contiguous().to(dev), t[:5].contiguous().to(dev)
prog, names = program_from_spec(spec, sd, dev)
prog.set_deterministic(mode != "nondet")
pd = pde_desc_from_spec(pde)
import collections
stat = collections.Counter()
for it in range(300):
    flat = E.new_flat_grad(prog, dev)
    flat.fill_(5.0)
    for wsb in E._workspaces.values(): wsb.view(torch.float32).fill_(float("nan"))
    E.residual_loss_grad(prog, pd, x, t, 1.0 / 5, flat)
    by = dict(zip(names, E.split_flat_grad(prog, flat.cpu())))
    for k in ("model.layers.0.bias", "model.layers.1.bias", "model.layers.2.bias", "model.layers.1.weight"):
        stat[(k, round(float(by[k].flatten()[0]), 6))] += 1
for k, v in sorted(stat.items()): print(k, v)
