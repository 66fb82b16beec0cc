"""Host-side checks of the C ABI against the sanitizer build (make -C pinns-rl-pde_amd/csrc asan), no GPU, no torch:

    LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
        python tools/asan_host_checks.py pinns-rl-pde_amd/libpinnjet_asan.so

Drives descriptor validation, program building and workspace sizing for every architecture at its deepest supported
size, and the table-length checks with deliberately SHORT tables (heap-allocated to their exact size, so that any read
past the end is an AddressSanitizer report; round 1 indexed a stack table on trust and overran it).
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pinnrl_amd  # noqa: E402,F401  (does not import torch)
from pinnrl_amd import _lib  # noqa: E402

lib = ctypes.CDLL(sys.argv[1])
P, vp, i32, i64 = ctypes.POINTER, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.pinn_last_error.restype = ctypes.c_char_p
lib.pinn_build_info.restype = ctypes.c_char_p
lib.pinn_workspace_bytes.restype = ctypes.c_size_t
lib.pinn_workspace_bytes.argtypes = [P(_lib.PinnNetDesc), i64, i32, i32, i32]
lib.pinn_num_tensors.argtypes = [P(_lib.PinnNetDesc)]
lib.pinn_jet_forward.argtypes = [P(_lib.PinnNetDesc), P(vp), i32, vp, vp, i64, i32, i32, P(vp), vp, ctypes.c_size_t, vp]
lib.pinn_residual_loss_grad.argtypes = [P(_lib.PinnNetDesc), P(vp), i32, P(_lib.PinnPdeDesc), vp, vp, i64, ctypes.c_float, vp, vp,
                                        P(vp), vp, ctypes.c_size_t, vp]
assert b"sanitizer" in lib.pinn_build_info()


def desc(arch, widths, input_dim=2, act="tanh", mapping=0, blocks=0, flags=0):
    d = _lib.PinnNetDesc()
    d.arch, d.activation, d.input_dim, d.num_linear = _lib.ARCH[arch], _lib.ACT[act], input_dim, len(widths)
    for i, w in enumerate(widths):
        d.widths[i] = w
    d.mapping_size, d.num_blocks, d.flags, d.ln_eps = mapping, blocks, flags, 1e-5
    return d


cases = {
    "fourier 4x128": (desc("fourier", [128, 128, 128, 1], mapping=32), 9),
    "feedforward 23 hidden layers x 1024 + LayerNorm": (desc("feedforward", [1024] * 23 + [1], flags=1), 94),
    "feedforward widths 50/70/33": (desc("feedforward", [50, 70, 33, 1]), 8),
    "siren 8x256": (desc("siren", [256] * 8 + [1], act="sin"), 18),
    "resnet 11 blocks x 512": (desc("resnet", [512] * 23 + [1], blocks=11), 92),
    "attention 6 layers x 256": (desc("attention", [256, 1], input_dim=3, act="gelu", blocks=6), 100),
}
for name, (d, want) in cases.items():
    n = lib.pinn_num_tensors(ctypes.byref(d))
    assert n == want, (name, n, lib.pinn_last_error())
    for nt, nx in ((0, 0), (1, 0), (1, 2), (1, 4), (2, 2)):
        for bwd in (0, 1):
            for N in (1, 31, 32, 33, 49729, 1000000):
                b = lib.pinn_workspace_bytes(ctypes.byref(d), N, nt, nx, bwd)
                assert b % 16 == 0
        d.flags |= 2
        assert lib.pinn_workspace_bytes(ctypes.byref(d), 5000, nt, nx, 1) >= lib.pinn_workspace_bytes(ctypes.byref(d), 5000, nt, nx, 0)
        d.flags &= ~2
    # tables of the wrong length are refused before any entry is read; heap tables of exactly that (short) size
    for bad in (0, 1, want - 1, want + 1):
        tbl = (vp * max(bad, 1))()
        outs = (vp * 7)()
        rc = lib.pinn_jet_forward(ctypes.byref(d), tbl, bad, 16, 16, 100, 1, 0, outs, None, 0, None)
        assert rc == -1, (name, bad, rc)
        pd = _lib.PinnPdeDesc()
        rc = lib.pinn_residual_loss_grad(ctypes.byref(d), tbl, bad, ctypes.byref(pd), 16, 16, 100, 1.0, None, None, tbl, None, 0, None)
        assert rc == -1, (name, bad, rc)
    print(f"ok  {name}: {n} tensors")
for bad_desc in (desc("fourier", [2000, 1], mapping=32), desc("resnet", [64, 64, 1], blocks=5), desc("feedforward", [64, 2])):
    assert lib.pinn_num_tensors(ctypes.byref(bad_desc)) < 0 and lib.pinn_workspace_bytes(ctypes.byref(bad_desc), 100, 1, 2, 1) == 0
print("host checks passed under the sanitizer build")
