"""pinnrl_amd — MI355X-native engine for pinnrl's collocation-point hot path.

Drop-in for `PINNModel` forward + `XxxEquation.compute_residual` + the loss
gradient, behind the reference's own Python API.  All
arithmetic runs in hand-written HIP kernels (`csrc/`, C ABI in
`include/pinn_jet.h`); PyTorch owns memory, streams and `torch.distributed`.
There is no CPU or eager fallback: without the HIP library or a ROCm device the
compute entry points raise.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401
