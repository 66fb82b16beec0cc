"""Where does a tile spend its cycles?  Needs the diagnostic library:
    make -C pinns-rl-pde_amd/csrc dev STAMPS=1 && PINN_LIB=pinns-rl-pde_amd/libpinnjet_dev.so python tools/stamps.py
Prints per-phase shares of wave lifetime (shader cycles from s_memtime), median over waves."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import _burgers
from pinnrl_amd import engine as E, _lib
import oracle as O

NAMES = ["stage", "encode", "fwd_gemm", "fwd_ew", "out", "epi", "b0", "bwd_ew", "bwd_stream", "bwd_flush", "enc_bwd", "TOTAL", "bwd_dx", "bwd_wait"]
dev = torch.device("cuda:0")
# usage: stamps.py [N]  (headline network)   |   stamps.py C3|C4|C5  (a BASELINE configuration from bench_configs.py)
if len(sys.argv) > 1 and sys.argv[1].startswith("C"):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as B
    name, model, pde, n_req = B.CONFIGS[sys.argv[1]]()
    torch.manual_seed(1)
    x, t = pde.generate_collocation_points(n_req, strategy="uniform")
else:
    cfg, model, pde = _burgers(dev, hidden=128, layers=4, mapping=32, scale=10.0)
    torch.manual_seed(1)
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    x, t = O.sample_uniform(O.PdeSpec(name="burgers"), n_req)
    x, t = x.to(dev), t.to(dev)
prog, pd = model.program(), pde._pde_desc()
flat = E.new_flat_grad(prog, dev)
lib = _lib.load()
buf = torch.zeros(1024 * 4 * 16, dtype=torch.int64, device=dev)
lib.pinn_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.pinn_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3):
    E.residual_loss_grad(prog, pd, x, t, 1.0 / x.shape[0], flat)
torch.cuda.synchronize()
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
buf.zero_()
s0.record(); E.residual_loss_grad(prog, pd, x, t, 1.0 / x.shape[0], flat); s1.record()
torch.cuda.synchronize()
st = buf.view(-1, 4, 16).cpu().double()
live = st[:, :, 11].sum(1) > 0
st = st[live]
tot = st[:, :, 11]
print(f"N={x.shape[0]} kernel {s0.elapsed_time(s1):.3f} ms; workgroups with work {int(live.sum())}; wave lifetime median {tot.median():.0f} cycles")
for i, nm in enumerate(NAMES):
    if nm == "TOTAL":
        continue
    v = st[:, :, i]
    print(f"  {nm:11s} median {v.median():10.0f}  share {100 * v.sum() / tot.sum():5.1f}%   (wave0 {st[:, 0, i].median():9.0f}  wave3 {st[:, 3, i].median():9.0f})")
