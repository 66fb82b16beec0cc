"""Tensor-level front end of the HIP jet engine.

`NetProgram` describes one network as the C ABI wants it (descriptor + the live
parameter tensors in `state_dict` order); the functions below launch the fused
kernels on the current HIP stream with raw device pointers, and the two
`autograd.Function`s splice them into PyTorch graphs so that the reference's
call pattern (`residual = pde.compute_residual(...)`, `loss.backward()`,
`optimizer.step()`) keeps working unchanged.

Everything here requires a ROCm device and the compiled library; nothing falls
back to eager PyTorch.
"""

from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


def _require_device(*tensors: Tensor) -> torch.device:
    dev = None
    for x in tensors:
        if x is None:
            continue
        if not x.is_cuda:
            raise RuntimeError(
                "pinnrl_amd: the HIP jet engine only runs on a ROCm device (got a CPU tensor); there is no CPU fallback"
            )
        dev = dev or x.device
        if x.device != dev:
            raise RuntimeError(f"pinnrl_amd: tensors on different devices ({x.device} vs {dev})")
    return dev


def _f32c(x: Tensor) -> Tensor:
    if x.dtype != torch.float32:
        x = x.float()
    return x if x.is_contiguous() else x.contiguous()


class NetProgram:
    """One network in the form the C ABI consumes.

    tensors: parameters AND buffers in the reference's `state_dict` order (fourier: B first).
    trainable[i] is False for buffers (no gradient slot).
    """

    def __init__(self, arch: str, activation: str, input_dim: int, widths: Sequence[int], tensors: Sequence[Tensor],
                 trainable: Sequence[bool], mapping_size: int = 0, omega_0: float = 0.0, ln_eps: float = 1e-5,
                 num_blocks: int = 0, layer_norm: bool = False, deterministic: bool = False):
        if arch not in _lib.ARCH:
            raise NotImplementedError(f"pinnrl_amd: architecture '{arch}' has no fused HIP kernel")
        if activation not in _lib.ACT:
            raise ValueError(f"Unsupported activation: {activation}")  # base_network.py:104
        if len(widths) > _lib.PINN_MAX_LINEAR:
            raise NotImplementedError(f"pinnrl_amd: more than {_lib.PINN_MAX_LINEAR} Linear layers")
        d = _lib.PinnNetDesc()
        d.arch = _lib.ARCH[arch]
        d.activation = _lib.ACT[activation]
        d.input_dim = int(input_dim)
        d.num_linear = len(widths)
        for i, w in enumerate(widths):
            d.widths[i] = int(w)
        d.mapping_size = int(mapping_size)
        d.act_param = float(omega_0)
        d.ln_eps = float(ln_eps)
        d.num_blocks = int(num_blocks)
        d.flags = (_lib.PINN_FLAG_LAYER_NORM if layer_norm else 0) | (_lib.PINN_FLAG_DETERMINISTIC if deterministic else 0)
        self.desc = d
        self.arch = arch
        self.tensors = list(tensors)
        self.trainable = list(trainable)
        self.input_dim = int(input_dim)

    def set_deterministic(self, on: bool = True) -> None:
        """Weight gradients reduced in a fixed order (bit-identical across launches on the same inputs)."""
        if on:
            self.desc.flags |= _lib.PINN_FLAG_DETERMINISTIC
        else:
            self.desc.flags &= ~_lib.PINN_FLAG_DETERMINISTIC

    def set_layer_major(self, on: bool = True) -> None:
        """Engine hint: take the layer-major engine even where the fused tile-major kernel applies."""
        if on:
            self.desc.flags |= _lib.PINN_FLAG_LAYER_MAJOR
        else:
            self.desc.flags &= ~_lib.PINN_FLAG_LAYER_MAJOR

    # -- pointer tables ---------------------------------------------------------------------
    @property
    def num_tensors(self) -> int:
        return len(self.tensors)

    def _weight_ptrs(self):
        for p in self.tensors:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("pinnrl_amd: parameters must be contiguous float32")
        return (ctypes.c_void_p * len(self.tensors))(*[p.data_ptr() for p in self.tensors])

    def grad_layout(self) -> Tuple[List[int], int]:
        """Offsets (in floats, 16-byte aligned) of every trainable tensor inside one flat gradient buffer."""
        offs, n = [], 0
        for p, tr in zip(self.tensors, self.trainable):
            offs.append(n if tr else -1)
            if tr:
                n += (p.numel() + 3) // 4 * 4
        return offs, n

    def flops_per_point(self) -> int:
        """F_fwd = 2 * sum(in*out) over Linear layers (+ the Fourier projection) — SURVEY.md §8(d)."""
        d = self.desc
        widths = [d.widths[i] for i in range(d.num_linear)]
        if self.arch == "fourier":
            prev, total = 2 * d.mapping_size, 2 * d.input_dim * d.mapping_size
        elif self.arch == "resnet":
            prev, total = d.input_dim, 0  # widths = [H] * (1 + 2 * blocks) + [out]: every Linear is listed
        elif self.arch == "attention":  # live Linears only: in, per layer value + proj + 2 x (H x 4H), out
            H = widths[0]
            return 2 * (d.input_dim * H + d.num_blocks * (2 * H * H + 8 * H * H) + H * widths[-1])
        else:
            prev, total = d.input_dim, 0
        for w in widths:
            total += 2 * prev * w
            prev = w
        return total


_workspaces: Dict[Tuple[torch.device, int], Tensor] = {}
_graph_pinned: Dict[Tuple[torch.device, int], bool] = {}
_retired: List[Tensor] = []  # buffers a captured HIP graph still points into: kept alive for the life of the process


def _workspace(dev: torch.device, nbytes: int) -> Tensor:
    """Reusable scratch owned by PyTorch's caching allocator (the library never allocates), one per (device, stream).

    A HIP graph bakes the buffer's address into its kernel arguments, so a buffer that was handed out during a
    capture is never returned to the allocator: when a later, larger request replaces it, it is parked in
    `_retired` instead of being freed (a freed block could be re-issued and the graph's replays would write over it)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and _graph_pinned.get(key):
            _retired.append(ws)
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
        _graph_pinned[key] = False
    if torch.cuda.is_current_stream_capturing():
        _graph_pinned[key] = True
    return ws


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _scratch(prog: "NetProgram", dev: torch.device, N: int, nt: int, nx: int, backward: bool):
    """(tensor | None, data_ptr, nbytes) of the scratch one call needs (records of the layer-major engine / tape)."""
    nbytes = _lib.load().pinn_workspace_bytes(ctypes.byref(prog.desc), N, nt, nx, 1 if backward else 0)
    if nbytes == 0:
        return None, None, 0
    ws = _workspace(dev, nbytes)
    return ws, ws.data_ptr(), ws.numel()


def _prep_points(prog: NetProgram, x: Tensor, t: Tensor) -> Tuple[Tensor, Tensor, int]:
    x, t = _f32c(x.detach()), _f32c(t.detach())
    if x.dim() != 2 or t.dim() != 2 or t.shape[1] != 1 or x.shape[0] != t.shape[0]:
        raise ValueError(f"expected x:(N,dim), t:(N,1); got {tuple(x.shape)}, {tuple(t.shape)}")
    if x.shape[1] + 1 != prog.input_dim:
        raise ValueError(f"model input_dim={prog.input_dim} but cat([x,t]) has {x.shape[1] + 1} columns")
    return x, t, x.shape[0]


def pde_desc(kind: str, dimension: int = 1, coef: Sequence[float] = (), loss: str = "mse", huber_delta: float = 1.0):
    d = _lib.PinnPdeDesc()
    d.kind = _lib.PDE[kind]
    d.dimension = int(dimension)
    d.loss = _lib.LOSS.get(loss, 0)  # unknown names fall back to mse (pde_base.py:313-315)
    for i, c in enumerate(coef):
        d.coef[i] = float(c)
    d.huber_delta = float(huber_delta)
    return d


def pde_streams(pd) -> Tuple[int, int]:
    nt, nx = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(_lib.load().pinn_pde_streams(ctypes.byref(pd), ctypes.byref(nt), ctypes.byref(nx)))
    return nt.value, nx.value


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
def jets_forward(prog: NetProgram, x: Tensor, t: Tensor, nt: int, nx: int) -> Tensor:
    """(K, N) tensor of [u, d/dt.., d/dx..] — one launch, no graph."""
    lib = _lib.load()
    dev = _require_device(x, t, *prog.tensors)
    x, t, N = _prep_points(prog, x, t)
    K = 1 + nt + nx
    out = torch.empty((K, N), dtype=torch.float32, device=dev)
    if N == 0:
        return out
    optr = (ctypes.c_void_p * K)(*[out[s].data_ptr() for s in range(K)])
    ws, wptr, wn = _scratch(prog, dev, N, nt, nx, False)
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_jet_forward(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors, x.data_ptr(),
                                        t.data_ptr(), N, nt, nx, optr, wptr, wn, _stream(dev)))
    return out


def _grad_ptrs(prog: NetProgram, flat: Tensor):
    offs, _ = prog.grad_layout()
    base = flat.data_ptr()
    return (ctypes.c_void_p * len(offs))(*[(base + 4 * o) if o >= 0 else None for o in offs])


def new_flat_grad(prog: NetProgram, dev: torch.device) -> Tensor:
    _, n = prog.grad_layout()
    return torch.zeros(n, dtype=torch.float32, device=dev)


def split_flat_grad(prog: NetProgram, flat: Tensor) -> List[Optional[Tensor]]:
    offs, _ = prog.grad_layout()
    return [flat[o : o + p.numel()].view_as(p) if o >= 0 else None for o, p in zip(offs, prog.tensors)]


def jets_backward(prog: NetProgram, x: Tensor, t: Tensor, nt: int, nx: int, cot: Tensor, flat_grad: Tensor) -> None:
    """flat_grad += d<cot, jets>/d(theta).  cot: (K, N)."""
    lib = _lib.load()
    dev = _require_device(x, t, cot, flat_grad, *prog.tensors)
    x, t, N = _prep_points(prog, x, t)
    if N == 0:
        return
    K = 1 + nt + nx
    cot = _f32c(cot)
    assert cot.shape == (K, N)
    cptr = (ctypes.c_void_p * K)(*[cot[s].data_ptr() for s in range(K)])
    ws, wptr, wn = _scratch(prog, dev, N, nt, nx, True)
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_jet_backward(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors, x.data_ptr(),
                                         t.data_ptr(), N, nt, nx, cptr, _grad_ptrs(prog, flat_grad), wptr, wn, _stream(dev)))


def residual_forward(prog: NetProgram, pd, x: Tensor, t: Tensor, want_residual: bool = True) -> Tuple[Optional[Tensor], Tensor]:
    """(residual (N,1) | None, loss_sum (1,)) with loss_sum = sum_n l(r_n) over THESE points."""
    lib = _lib.load()
    dev = _require_device(x, t, *prog.tensors)
    x, t, N = _prep_points(prog, x, t)
    r = torch.empty((N, 1), dtype=torch.float32, device=dev) if want_residual else None
    s = torch.zeros(1, dtype=torch.float32, device=dev)
    if N:
        nt, nx = pde_streams(pd)
        ws, wptr, wn = _scratch(prog, dev, N, nt, nx, False)
        with torch.cuda.device(dev):
            _lib.check(lib.pinn_residual_forward(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors,
                                                 ctypes.byref(pd), x.data_ptr(), t.data_ptr(), N,
                                                 r.data_ptr() if want_residual else None, s.data_ptr(), wptr, wn,
                                                 _stream(dev)))
    return r, s


def residual_loss_grad(prog: NetProgram, pd, x: Tensor, t: Tensor, grad_scale: float, flat_grad: Tensor,
                       want_residual: bool = False, loss_sum: Optional[Tensor] = None,
                       coef_grads: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Tensor]:
    """One launch: flat_grad += grad_scale * d(sum_n l(r_n))/d(theta); returns (residual | None, loss_sum).
    `coef_grads` (>= 2 floats on the device, accumulated): the same derivative w.r.t. the PDE coefficients pd.coef[0..1]
    (inverse problems), from the same launch."""
    lib = _lib.load()
    dev = _require_device(x, t, flat_grad, *prog.tensors)
    x, t, N = _prep_points(prog, x, t)
    r = torch.empty((N, 1), dtype=torch.float32, device=dev) if want_residual else None
    s = loss_sum if loss_sum is not None else torch.zeros(1, dtype=torch.float32, device=dev)
    if N:
        nt, nx = pde_streams(pd)
        flags0 = prog.desc.flags
        if coef_grads is not None:  # the coefficient reduction lives in the layer-major engine (sizing and call alike)
            prog.desc.flags = flags0 | _lib.PINN_FLAG_LAYER_MAJOR
        try:
            ws, wptr, wn = _scratch(prog, dev, N, nt, nx, True)
        except Exception:
            prog.desc.flags = flags0
            raise
        with torch.cuda.device(dev):
            if coef_grads is not None:
                assert coef_grads.is_cuda and coef_grads.dtype == torch.float32 and coef_grads.numel() >= 2
                try:
                    _lib.check(lib.pinn_residual_loss_grad_coef(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors,
                                                                ctypes.byref(pd), x.data_ptr(), t.data_ptr(), N, float(grad_scale),
                                                                r.data_ptr() if want_residual else None, s.data_ptr(),
                                                                _grad_ptrs(prog, flat_grad), coef_grads.data_ptr(), wptr, wn,
                                                                _stream(dev)))
                finally:
                    prog.desc.flags = flags0
            else:
                _lib.check(lib.pinn_residual_loss_grad(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors,
                                                       ctypes.byref(pd), x.data_ptr(), t.data_ptr(), N, float(grad_scale),
                                                       r.data_ptr() if want_residual else None, s.data_ptr(),
                                                       _grad_ptrs(prog, flat_grad), wptr, wn, _stream(dev)))
    return r, s


def residual_backward(prog: NetProgram, pd, x: Tensor, t: Tensor, res_bar: Tensor, flat_grad: Tensor) -> None:
    """flat_grad += d<res_bar, r>/d(theta) (forward recomputed per tile inside the launch)."""
    lib = _lib.load()
    dev = _require_device(x, t, res_bar, flat_grad, *prog.tensors)
    x, t, N = _prep_points(prog, x, t)
    if N == 0:
        return
    res_bar = _f32c(res_bar).reshape(-1)
    assert res_bar.numel() == N
    nt, nx = pde_streams(pd)
    ws, wptr, wn = _scratch(prog, dev, N, nt, nx, True)
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_residual_backward(ctypes.byref(prog.desc), prog._weight_ptrs(), prog.num_tensors,
                                              ctypes.byref(pd), x.data_ptr(), t.data_ptr(), N, res_bar.data_ptr(),
                                              _grad_ptrs(prog, flat_grad), wptr, wn, _stream(dev)))


# ---------------------------------------------------------------------------------------------
# training-step kernels (no autograd): point-wise loss terms, clip + Adam over a flat buffer
# ---------------------------------------------------------------------------------------------
def point_losses(u: Tensor, terms: Sequence[Tuple[int, int, Tensor, float]], loss: str, huber_delta: float,
                 term_losses: Tensor, cot: Tensor, residual_sum: Optional[Tensor] = None, residual_scale: float = 0.0,
                 residual_weight: float = 0.0, n_boundary_terms: int = 0, summary4: Optional[Tensor] = None) -> None:
    """term k = (lo, hi, target (hi - lo,), weight): term_losses[k] = mean l(u[lo:hi] - target);
    cot[n] = sum_k weight_k l'(.) / (hi - lo); summary4 = {residual, boundary, initial, total}.  One launch."""
    lib = _lib.load()
    dev = _require_device(u, term_losses, cot, *[t[2] for t in terms])
    n = len(terms)
    lo = (ctypes.c_int32 * n)(*[int(t[0]) for t in terms])
    hi = (ctypes.c_int32 * n)(*[int(t[1]) for t in terms])
    tg = (ctypes.c_void_p * n)(*[t[2].data_ptr() for t in terms])
    w = (ctypes.c_float * n)(*[float(t[3]) for t in terms])
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_point_losses(u.data_ptr(), u.numel(), n, lo, hi, tg, w, _lib.LOSS.get(loss, 0),
                                         float(huber_delta), term_losses.data_ptr(), cot.data_ptr(),
                                         residual_sum.data_ptr() if residual_sum is not None else None,
                                         float(residual_scale), float(residual_weight), int(n_boundary_terms),
                                         summary4.data_ptr() if summary4 is not None else None, _stream(dev)))


def jet_losses(jets: Tensor, terms: Sequence[Tuple[int, int, int, int, Optional[Tensor], float]], loss: str, huber_delta: float,
               term_losses: Tensor, cot: Tensor, residual_sum: Optional[Tensor] = None, residual_scale: float = 0.0,
               residual_weight: float = 0.0, n_boundary_terms: int = 0, summary4: Optional[Tensor] = None) -> None:
    """General form of `point_losses` on (K, n) jets.  term k = (lo, hi, stream, pair_offset, target | None, weight):
    mean l(J[stream, lo:hi] - target) or, with pair_offset != 0, mean l(J[stream, lo:hi] - J[stream, lo+pair : hi+pair])
    (periodic boundary pairs); cot (K, n) is overwritten with the cotangents of the jets.  One launch."""
    lib = _lib.load()
    dev = _require_device(jets, term_losses, cot, *[t[4] for t in terms if t[4] is not None])
    assert jets.dim() == 2 and jets.is_contiguous() and cot.shape == jets.shape and cot.is_contiguous()
    n = len(terms)
    lo = (ctypes.c_int32 * n)(*[int(t[0]) for t in terms])
    hi = (ctypes.c_int32 * n)(*[int(t[1]) for t in terms])
    st = (ctypes.c_int32 * n)(*[int(t[2]) for t in terms])
    pr = (ctypes.c_int32 * n)(*[int(t[3]) for t in terms])
    tg = (ctypes.c_void_p * n)(*[t[4].data_ptr() if t[4] is not None else None for t in terms])
    w = (ctypes.c_float * n)(*[float(t[5]) for t in terms])
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_jet_losses(jets.data_ptr(), jets.shape[0], jets.shape[1], n, lo, hi, st, pr, tg, w, _lib.LOSS.get(loss, 0),
                                       float(huber_delta), term_losses.data_ptr(), cot.data_ptr(),
                                       residual_sum.data_ptr() if residual_sum is not None else None, float(residual_scale),
                                       float(residual_weight), int(n_boundary_terms),
                                       summary4.data_ptr() if summary4 is not None else None, _stream(dev)))


def adam_clip_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: Tensor, step: Tensor,
                   scratch: Tensor, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                   weight_decay: float = 0.0, max_norm: float = 0.0, grad_norm_out: Optional[Tensor] = None) -> None:
    """clip_grad_norm_(max_norm) + Adam on flat fp32 buffers; `lr` and `step` are device scalars.  Three tiny launches."""
    lib = _lib.load()
    dev = _require_device(params, grads, exp_avg, exp_avg_sq, lr, step, scratch)
    n = params.numel()
    assert grads.numel() >= n and exp_avg.numel() == n and exp_avg_sq.numel() == n and scratch.numel() >= 64
    with torch.cuda.device(dev):
        _lib.check(lib.pinn_adam_clip_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                           lr.data_ptr(), beta1, beta2, eps, weight_decay, max_norm, step.data_ptr(),
                                           scratch.data_ptr(), grad_norm_out.data_ptr() if grad_norm_out is not None else None,
                                           _stream(dev)))


# ---------------------------------------------------------------------------------------------
# autograd splices
# ---------------------------------------------------------------------------------------------
class JetFunction(torch.autograd.Function):
    """jets = f(theta; x, t), differentiable w.r.t. the trainable tensors of the program.

    Replaces `u = model(cat[x,t])` + the chained `autograd.grad(create_graph=True)` calls of
    `PDEBase.compute_derivatives` (pinnrl/pdes/pde_base.py:640-732).  The coordinates are treated
    as constants, as the reference does after its `detach()` (pde_base.py:630-631).
    """

    @staticmethod
    def forward(ctx, prog: NetProgram, x: Tensor, t: Tensor, nt: int, nx: int, *params: Tensor):
        ctx.prog, ctx.nt, ctx.nx = prog, nt, nx
        ctx.save_for_backward(x, t)
        return jets_forward(prog, x, t, nt, nx)

    @staticmethod
    def backward(ctx, cot: Tensor):
        prog = ctx.prog
        x, t = ctx.saved_tensors
        flat = new_flat_grad(prog, cot.device)
        jets_backward(prog, x, t, ctx.nt, ctx.nx, cot, flat)
        grads = split_flat_grad(prog, flat)
        return (None, None, None, None, None, *grads)


class ResidualFunction(torch.autograd.Function):
    """r = residual(theta; x, t) as an (N, 1) tensor with a grad_fn — `XxxEquation.compute_residual`."""

    @staticmethod
    def forward(ctx, prog: NetProgram, pd, x: Tensor, t: Tensor, *params: Tensor):
        ctx.prog, ctx.pd = prog, pd
        ctx.save_for_backward(x, t)
        r, _ = residual_forward(prog, pd, x, t, want_residual=True)
        return r

    @staticmethod
    def backward(ctx, rbar: Tensor):
        x, t = ctx.saved_tensors
        flat = new_flat_grad(ctx.prog, rbar.device)
        residual_backward(ctx.prog, ctx.pd, x, t, rbar, flat)
        return (None, None, None, None, *split_flat_grad(ctx.prog, flat))


class ResidualLossFunction(torch.autograd.Function):
    """mean_n l(r_n) with the gradient produced by the SAME launch (fused forward + reverse sweep).

    Replaces `residual = compute_residual(...)`, `_apply_loss_fn(residual)` and the residual branch of
    `loss.backward()` (pde_base.py:1098-1099, trainer.py:689).  `n_total` is the global point count
    when the batch is sharded over ranks (the local sum is divided by it).
    """

    @staticmethod
    def forward(ctx, prog: NetProgram, pd, x: Tensor, t: Tensor, n_total: int, *params: Tensor):
        need_grad = any(ctx.needs_input_grad[5:])  # grad mode is off inside forward(); this reflects the caller's
        dev = x.device
        if need_grad:
            flat = new_flat_grad(prog, dev)
            _, s = residual_loss_grad(prog, pd, x, t, 1.0 / float(n_total), flat)
            ctx.flat, ctx.prog = flat, prog
        else:
            _, s = residual_forward(prog, pd, x, t, want_residual=False)
            ctx.flat, ctx.prog = None, prog
        return (s / float(n_total)).reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        if ctx.flat is None:
            return (None,) * 5 + (None,) * len(ctx.prog.tensors)
        grads = split_flat_grad(ctx.prog, ctx.flat * g)
        return (None, None, None, None, None, *grads)


class ResidualLossCoefFunction(torch.autograd.Function):
    """`ResidualLossFunction` for inverse problems: the PDE coefficients `coefs` (tensors in the order of PinnPdeDesc.coef,
    some of which require grad: live `nn.Parameter`s of `PDEBase._trainable_params`, pde_base.py:246-279) are inputs of the
    node, and their gradients come from the SAME launch as the weight gradient (fused `sum_n rbar_n dr_n/dc_k` reduction in the
    residual epilogue) — no jets + torch epilogue, no second pass."""

    @staticmethod
    def forward(ctx, prog: NetProgram, make_pd, x: Tensor, t: Tensor, n_total: int, n_coef: int, *coefs_and_params: Tensor):
        coefs = coefs_and_params[:n_coef]
        pd = make_pd([float(c.detach()) for c in coefs])
        dev = x.device
        flat = new_flat_grad(prog, dev)
        cg = torch.zeros(4, dtype=torch.float32, device=dev)
        _, s = residual_loss_grad(prog, pd, x, t, 1.0 / float(n_total), flat, coef_grads=cg)
        ctx.flat, ctx.cg, ctx.prog, ctx.n_coef = flat, cg, prog, n_coef
        ctx.coef_meta = [(c.shape, c.dtype) for c in coefs]
        return (s / float(n_total)).reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        grads = split_flat_grad(ctx.prog, ctx.flat * g)
        cgs = []
        for k, (shape, dtype) in enumerate(ctx.coef_meta):
            cgs.append((ctx.cg[k] * g).to(dtype).reshape(shape) if k < 2 and ctx.needs_input_grad[6 + k] else None)
        return (None, None, None, None, None, None, *cgs, *grads)
