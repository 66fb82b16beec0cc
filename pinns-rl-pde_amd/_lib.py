"""ctypes binding of libpinnjet.so (C ABI: include/pinn_jet.h).  Fails loudly when the library is absent."""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PINN_LIB") or os.path.join(_HERE, "libpinnjet.so")  # PINN_LIB: developer builds
CSRC = os.path.join(_HERE, "csrc")

PINN_ABI_VERSION = 2
PINN_MAX_LINEAR = 24
PINN_MAX_STREAMS = 7
PINN_FLAG_LAYER_NORM = 1
PINN_FLAG_DETERMINISTIC = 2
PINN_FLAG_LAYER_MAJOR = 4

ARCH = {"feedforward": 0, "fourier": 1, "siren": 2, "resnet": 3, "attention": 4}
ACT = {"tanh": 0, "sin": 1, "gelu": 2, "sigmoid": 3, "relu": 4, "leaky_relu": 5, "identity": 6}
PDE = {
    "burgers": 0, "heat": 1, "allen_cahn": 2, "kdv": 3, "cahn_hilliard": 4, "wave": 5, "convection": 6,
    "black_scholes": 7, "pendulum": 8, "heat_laplacian": 9,
}
LOSS = {"mse": 0, "mae": 1, "huber": 2}

EXPORTS = (
    "pinn_abi_version", "pinn_last_error", "pinn_build_info", "pinn_num_tensors", "pinn_pde_streams",
    "pinn_workspace_bytes", "pinn_jet_forward", "pinn_jet_backward", "pinn_residual_forward", "pinn_residual_backward",
    "pinn_residual_loss_grad", "pinn_residual_loss_grad_coef", "pinn_point_losses", "pinn_jet_losses", "pinn_adam_clip_step",
)


class PinnNetDesc(ctypes.Structure):
    _fields_ = [
        ("arch", ctypes.c_int32), ("activation", ctypes.c_int32), ("input_dim", ctypes.c_int32),
        ("num_linear", ctypes.c_int32), ("widths", ctypes.c_int32 * PINN_MAX_LINEAR),
        ("mapping_size", ctypes.c_int32), ("act_param", ctypes.c_float), ("ln_eps", ctypes.c_float),
        ("num_blocks", ctypes.c_int32), ("flags", ctypes.c_int32),
    ]


class PinnPdeDesc(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32), ("dimension", ctypes.c_int32), ("loss", ctypes.c_int32),
        ("coef", ctypes.c_float * 4), ("huber_delta", ctypes.c_float),
    ]


class JetLibraryError(RuntimeError):
    """The HIP library is missing, stale, or returned an error code."""


_lib = None
_lock = threading.Lock()


def build(verbose: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU).  Returns the .so path."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise JetLibraryError(f"building libpinnjet.so failed (exit {res.returncode})")
    return LIB_PATH


def load():
    """dlopen libpinnjet.so once (after torch, so both share torch's HIP runtime) and type its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise JetLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C pinns-rl-pde_amd/csrc`).  There is no CPU fallback."
            )
        import torch  # noqa: F401  (loads libamdhip64 first; our .so binds to the same runtime by soname)

        lib = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        P = ctypes.POINTER
        lib.pinn_abi_version.restype = ctypes.c_int
        lib.pinn_abi_version.argtypes = []
        lib.pinn_last_error.restype = ctypes.c_char_p
        lib.pinn_last_error.argtypes = []
        lib.pinn_build_info.restype = ctypes.c_char_p
        lib.pinn_build_info.argtypes = []
        lib.pinn_num_tensors.restype = ctypes.c_int
        lib.pinn_num_tensors.argtypes = [P(PinnNetDesc)]
        lib.pinn_pde_streams.restype = ctypes.c_int
        lib.pinn_pde_streams.argtypes = [P(PinnPdeDesc), P(i32), P(i32)]
        lib.pinn_workspace_bytes.restype = ctypes.c_size_t
        lib.pinn_workspace_bytes.argtypes = [P(PinnNetDesc), i64, i32, i32, i32]
        sz = ctypes.c_size_t
        lib.pinn_jet_forward.restype = ctypes.c_int
        lib.pinn_jet_forward.argtypes = [P(PinnNetDesc), P(vp), i32, vp, vp, i64, i32, i32, P(vp), vp, sz, vp]
        lib.pinn_jet_backward.restype = ctypes.c_int
        lib.pinn_jet_backward.argtypes = [P(PinnNetDesc), P(vp), i32, vp, vp, i64, i32, i32, P(vp), P(vp), vp, sz, vp]
        lib.pinn_residual_forward.restype = ctypes.c_int
        lib.pinn_residual_forward.argtypes = [P(PinnNetDesc), P(vp), i32, P(PinnPdeDesc), vp, vp, i64, vp, vp, vp, sz, vp]
        lib.pinn_residual_backward.restype = ctypes.c_int
        lib.pinn_residual_backward.argtypes = [P(PinnNetDesc), P(vp), i32, P(PinnPdeDesc), vp, vp, i64, vp, P(vp), vp, sz,
                                               vp]
        lib.pinn_residual_loss_grad.restype = ctypes.c_int
        lib.pinn_residual_loss_grad.argtypes = [P(PinnNetDesc), P(vp), i32, P(PinnPdeDesc), vp, vp, i64, f32, vp, vp,
                                                P(vp), vp, sz, vp]
        lib.pinn_residual_loss_grad_coef.restype = ctypes.c_int
        lib.pinn_residual_loss_grad_coef.argtypes = [P(PinnNetDesc), P(vp), i32, P(PinnPdeDesc), vp, vp, i64, f32, vp, vp,
                                                     P(vp), vp, vp, sz, vp]
        lib.pinn_point_losses.restype = ctypes.c_int
        lib.pinn_point_losses.argtypes = [vp, i32, i32, P(i32), P(i32), P(vp), P(f32), i32, f32, vp, vp, vp, f32, f32, i32, vp, vp]
        lib.pinn_jet_losses.restype = ctypes.c_int
        lib.pinn_jet_losses.argtypes = [vp, i32, i32, i32, P(i32), P(i32), P(i32), P(i32), P(vp), P(f32), i32, f32, vp, vp, vp, f32, f32,
                                        i32, vp, vp]
        lib.pinn_adam_clip_step.restype = ctypes.c_int
        lib.pinn_adam_clip_step.argtypes = [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp, vp, vp, vp]
        if lib.pinn_abi_version() != PINN_ABI_VERSION:
            raise JetLibraryError(f"libpinnjet.so ABI {lib.pinn_abi_version()} != expected {PINN_ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


def build_info() -> str:
    """Kernel translation units that were built in a degraded form ('' when none), see csrc/Makefile."""
    return load().pinn_build_info().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        msg = load().pinn_last_error().decode("utf-8", "replace")
        if rc == -6:  # PINN_ERR_BAD_ORDER mirrors the reference's ValueError (pde_base.py:615-627)
            raise ValueError(msg)
        if rc == -2:
            raise NotImplementedError(f"pinn_jet: {msg}")
        raise JetLibraryError(f"pinn_jet error {rc}: {msg}")
