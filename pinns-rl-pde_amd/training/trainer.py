"""`PDETrainer` — counterpart of pinnrl/training/trainer.py for the accelerated path.

The step semantics of the reference's inner loop (trainer.py:539-698) are reproduced exactly:
`steps/epoch = num_points // batch_size`; fresh sample each step (strategy "adaptive" when an RL
agent is attached, else `config.training.collocation_distribution`); `zero_grad -> compute_loss ->
[adaptive weights] -> backward -> clip_grad_norm_ -> optimizer.step`; per-epoch scheduler step;
validation every `validation_frequency` epochs on 1000 (-> 961) points; early stopping on
`val < best - 1e-6`.  Like upstream, `optimizer_config` is accepted and ignored (lr / weight decay
come from `config.training`, trainer.py:292-297).  Plotting, JSON metadata, live snapshots and the
tqdm bar are dashboard plumbing and out of scope; `loss.item()` host syncs happen once per epoch
(plus `log_every_step=True` for reference-identical per-step bookkeeping).

Data-parallel (new): pass `process_group=` (one process per GPU).  Each rank trains
on its shard of every batch; ONE all-reduce per step carries [flat gradient || loss terms].
"""

from __future__ import annotations

import logging
from datetime import datetime
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import distributed as _D
from .. import engine as _E


class _EmaLossWeights:
    """The two EMA rules of pinnrl/components/adaptive_weights.py:31-111 (scalar bookkeeping on 3-4 numbers)."""

    def __init__(self, strategy="rbw", alpha=0.9, eps=1e-5, initial_weights=None):
        self.strategy, self.alpha, self.eps = strategy.lower(), alpha, float(eps)
        self.initial_weights = torch.tensor(initial_weights) if initial_weights is not None else None
        self.weights = self.running = self.prev_weights = None

    def update(self, losses=None, gradients=None):
        v = gradients if (self.strategy == "lrw" and gradients is not None) else losses
        if v is None or (self.strategy not in ("lrw", "rbw")):
            raise ValueError(f"Invalid combination of strategy ({self.strategy}) and inputs")
        if self.running is None:
            self.running = v
            self.weights = self.initial_weights.to(v.device) if self.initial_weights is not None else torch.ones_like(v)
            return self.weights
        self.running = self.alpha * self.running + (1 - self.alpha) * v
        if self.strategy == "lrw":
            inv = 1.0 / (self.running + self.eps)
            self.weights = inv / torch.sum(inv)
        else:
            self.weights = self.running / (self.running.sum() + self.eps)
            if self.prev_weights is not None:
                self.weights = self.alpha * self.prev_weights + (1 - self.alpha) * self.weights
            self.prev_weights = self.weights.clone()
        return self.weights


class PDETrainer:
    def __init__(self, model: nn.Module, pde, optimizer_config: Optional[Dict], config, device: Optional[torch.device] = None,
                 rl_agent=None, viz_frequency=10, validation_frequency=10, early_stopping_config=None,
                 process_group=None, log_every_step: bool = False, fast_step: Optional[bool] = None):
        self.device = device or (config.device if hasattr(config, "device") else torch.device("cpu"))
        self.model = model.to(self.device)
        self.pde = pde
        self.config = config
        self.validation_frequency = validation_frequency
        self.logger = logging.getLogger(__name__)
        self.process_group = process_group
        self.log_every_step = log_every_step
        # fast_step: None = `train()` takes the autograd-free launch list whenever it covers the configuration
        # (`_manual_step_unsupported()` is None) and the device is a GPU; False = always the autograd step; True = require it
        self.fast_step = fast_step
        if process_group is not None:
            # replicas must start from ONE theta_0 and apply identical updates: broadcast rank 0's parameters,
            # and refuse the modes whose per-rank quantities this path does not reduce
            if config.training.adaptive_weights.enabled:
                raise NotImplementedError("data-parallel training supports fixed loss weights only: the EMA weight rules "
                                          "need every loss component reduced over ranks before the update")
            _D.broadcast_parameters(self.model, process_group)
            if hasattr(self.pde, "_trainable_params") and len(self.pde._trainable_params):
                _D.broadcast_parameters(self.pde._trainable_params, process_group)
        self._initialize_optimizer_and_scheduler()
        self.history = {"train_loss": [], "val_loss": [], "residual_loss": [], "boundary_loss": [], "initial_loss": [],
                        "learning_rate": [], "loss_weights": []}
        if early_stopping_config is None:
            early_stopping_config = {"enabled": True, "patience": 10}
        self.early_stopping_enabled = early_stopping_config.get("enabled", True)
        self.patience = early_stopping_config.get("patience", 10)
        self.best_val_loss = float("inf")
        self.patience_counter = 0
        self.rl_agent = rl_agent
        self.viz_frequency = viz_frequency
        aw = config.training.adaptive_weights
        self.use_adaptive_weights = bool(aw.enabled)
        self.adaptive_weights = (
            _EmaLossWeights(aw.strategy, aw.alpha, aw.eps, aw.initial_weights) if self.use_adaptive_weights else None
        )
        self.points_history: List[np.ndarray] = []

    # ---------------------------------------------------------------- optimizer / scheduler (trainer.py:281-371)
    def _collect_optimizable_params(self):
        params = list(self.model.parameters())
        if hasattr(self.pde, "trainable_parameters_iter"):
            params += list(self.pde.trainable_parameters_iter())
        return params

    def _build_adam(self, params):
        tc = self.config.training
        # same update rule as the reference's optim.Adam(lr, weight_decay) (trainer.py:292-297); on the device the
        # multi-tensor update runs as ONE fused kernel instead of ~10 foreach launches
        fused = all(p.is_cuda and p.dtype == torch.float32 for p in params)
        return optim.Adam(params, lr=tc.learning_rate, weight_decay=tc.weight_decay, fused=fused)

    def _build_lbfgs(self, params):
        tc, c = self.config.training, self.config.training.lbfgs
        return optim.LBFGS(params, lr=tc.learning_rate, history_size=c.history_size, max_iter=c.max_iter,
                           line_search_fn=c.line_search_fn, tolerance_grad=c.tolerance_grad,
                           tolerance_change=c.tolerance_change)

    def _build_scheduler(self, force_reduce_lr: bool = False):
        sc = self.config.training.learning_rate_scheduler
        if force_reduce_lr or sc.type == "reduce_lr":
            return optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=sc.factor,
                                                        patience=sc.patience, min_lr=sc.min_lr)
        if sc.type == "cosine":
            return optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.config.training.num_epochs,
                                                        eta_min=sc.min_lr)
        raise ValueError(f"Unknown scheduler type: {sc.type}")

    def _initialize_optimizer_and_scheduler(self):
        kind = getattr(self.config.training, "optimizer", "adam")
        params = self._collect_optimizable_params()
        self._is_lbfgs = kind == "lbfgs"
        self.optimizer = self._build_lbfgs(params) if self._is_lbfgs else self._build_adam(params)
        if kind == "adam_lbfgs":
            ratio = getattr(self.config.training, "adam_lbfgs_switch_ratio", 0.7)
            self._switch_epoch = max(1, int(self.config.training.num_epochs * ratio))
        else:
            self._switch_epoch = None
        self.scheduler = self._build_scheduler(force_reduce_lr=self._is_lbfgs)
        self._optimizer_type = kind

    def _switch_to_lbfgs(self):
        self.optimizer = self._build_lbfgs(self._collect_optimizable_params())
        self._is_lbfgs = True
        self.scheduler = self._build_scheduler(force_reduce_lr=True)

    def _update_scheduler(self, val_loss=None):
        if isinstance(self.scheduler, optim.lr_scheduler.ReduceLROnPlateau):
            self.scheduler.step(val_loss)
        else:
            self.scheduler.step()

    # ---------------------------------------------------------------- validation (trainer.py:140-162)
    def _compute_validation_loss(self, num_points: int = 1000) -> Dict[str, float]:
        self.model.eval()
        x, t = self.pde.generate_collocation_points(num_points)
        with torch.no_grad():  # forward-only fused kernels; the reference builds (and discards) a graph here
            losses = self.pde.compute_loss(self.model, x.to(self.device), t.to(self.device))
        vals = torch.stack([losses[k].detach().reshape(()).float() for k in ("total", "residual", "boundary", "initial")])
        if self.process_group is not None:  # rank-local random validation points: average, so every rank stops together
            torch.distributed.all_reduce(vals, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
            vals = vals / torch.distributed.get_world_size(self.process_group)
        v = vals.tolist()
        return {"total_loss": v[0], "residual_loss": v[1], "boundary_loss": v[2], "initial_loss": v[3]}

    # ---------------------------------------------------------------- live snapshot (trainer.py:171-279)
    def live_snapshot_fields(self, grid_size: int = 60) -> Dict[str, object]:
        """The fields of the reference's `_save_live_snapshot` — predicted u and the PDE residual on a fixed
        grid_size x grid_size grid (x-t in 1-D; x1-x2 at the mid time in >= 2-D) — from two forward-only launches (no
        autograd graph; the reference builds one for the residual and throws it away).  Same keys as its `.npz`."""
        dev = self.device
        dim = int(getattr(self.pde, "dimension", 1))
        t_lo, t_hi = float(self.pde.time_domain[0]), float(self.pde.time_domain[1])
        if dim <= 1:
            xs = np.linspace(float(self.pde.domain[0][0]), float(self.pde.domain[0][1]), grid_size, dtype=np.float32)
            ys = np.linspace(t_lo, t_hi, grid_size, dtype=np.float32)
            xx, tt = np.meshgrid(xs, ys, indexing="xy")
            x_flat = torch.tensor(xx.reshape(-1, 1), device=dev)
            t_flat = torch.tensor(tt.reshape(-1, 1), device=dev)
            meta = {"dimension": 1, "x_label": "x", "y_label": "t", "fixed_t": float("nan")}
        else:
            xs = np.linspace(float(self.pde.domain[0][0]), float(self.pde.domain[0][1]), grid_size, dtype=np.float32)
            ys = np.linspace(float(self.pde.domain[1][0]), float(self.pde.domain[1][1]), grid_size, dtype=np.float32)
            xx1, xx2 = np.meshgrid(xs, ys, indexing="xy")
            cols = [xx1.reshape(-1), xx2.reshape(-1)] + [np.full(xx1.size, 0.5 * (float(self.pde.domain[d][0]) + float(self.pde.domain[d][1])),
                                                                 dtype=np.float32) for d in range(2, dim)]
            x_flat = torch.tensor(np.stack(cols, axis=1), device=dev, dtype=torch.float32)
            fixed_t = 0.5 * (t_lo + t_hi)
            t_flat = torch.full((x_flat.shape[0], 1), fixed_t, dtype=torch.float32, device=dev)
            meta = {"dimension": 2, "x_label": "x1", "y_label": "x2", "fixed_t": float(fixed_t)}
        was_training = self.model.training
        try:
            with torch.no_grad():
                u = self.model(torch.cat([x_flat, t_flat], dim=1))
                r = self.pde.compute_residual(self.model, x_flat, t_flat)  # forward-only kernel under no_grad
        finally:
            self.model.train(was_training)
        out = {"axis_x": xs, "axis_y": ys, "u_pred": u.detach().cpu().numpy()[:, 0].reshape(grid_size, grid_size),
               "residual": r.detach().cpu().numpy().reshape(grid_size, grid_size)}
        out.update(meta)
        return out

    def _save_live_snapshot(self, experiment_dir: str, epoch: int, grid_size: int = 60) -> None:
        if not experiment_dir:
            return
        try:
            import os

            np.savez(os.path.join(experiment_dir, "live_snapshot.npz"), epoch=int(epoch), **self.live_snapshot_fields(grid_size))
        except Exception as exc:  # viz must never crash the training loop (trainer.py:186-188)
            self.logger.debug(f"Live snapshot skipped: {exc}")

    # ---------------------------------------------------------------- one step
    def _sample(self, batch_size: int):
        strategy = "adaptive" if self.rl_agent is not None else self.config.training.collocation_distribution
        if strategy == "adaptive" and getattr(self, "_flat", None) is not None and hasattr(self.pde, "_sample_adaptive_device"):
            # the autograd-free step samples without a host round trip (device-side action selection and epsilon decay)
            if getattr(self.pde, "rl_agent", None) is None:
                self.pde.rl_agent = self.rl_agent
            return self.pde._sample_adaptive_device(batch_size)
        kw = {"model": self.model} if strategy == "residual_based" else {}
        x, t = self.pde.generate_collocation_points(batch_size, strategy=strategy, **kw)
        return x.to(self.device), t.to(self.device)

    def _lbfgs_step(self, x, t):  # trainer.py:373-389
        captured: Dict[str, Dict] = {}

        def closure():
            self.optimizer.zero_grad()
            losses = self._losses(x, t)
            losses["total"].backward()
            self._sync_grads(losses)
            captured["losses"] = losses
            return losses["total"]

        self.optimizer.step(closure)
        if "losses" not in captured:
            captured["losses"] = self._losses(x, t)
        return captured["losses"]

    def _losses(self, x, t):
        if self.process_group is not None:
            return _D.sharded_compute_loss(self.pde, self.model, x, t, self.process_group)
        return self.pde.compute_loss(self.model, x, t)

    def _ensure_dp_buffer(self):
        """Every parameter's `.grad` becomes a VIEW of one persistent [gradients || 4 scalars] buffer: backward
        accumulates straight into it and the step's single all-reduce runs on it in place (no cat / copy-back)."""
        if getattr(self, "_dp_buf", None) is not None:
            return
        params = self._collect_optimizable_params()
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        buf = torch.zeros(n + 4, dtype=torch.float32, device=params[0].device)
        for p, o in zip(params, offs):
            p.grad = buf[o : o + p.numel()].view_as(p)
        self._dp_buf, self._dp_n = buf, n

    def _sync_grads(self, losses=None):
        """ONE all-reduce of [gradients || residual loss]; the reduced (global) residual replaces the local shard's."""
        if self.process_group is None:
            return
        red = None
        if getattr(self, "_dp_buf", None) is not None:
            buf = self._dp_buf
            if losses is not None:
                buf[self._dp_n] = losses["residual"].detach().reshape(())
            torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
            red = buf[self._dp_n : self._dp_n + 1]
        else:
            scalars = [losses["residual"]] if losses is not None else None
            red = _D.all_reduce_gradients(self._collect_optimizable_params(), self.process_group, scalars=scalars)
        if red is not None and losses is not None:
            world = torch.distributed.get_world_size(self.process_group)
            local = losses["residual"].detach()
            losses["residual"] = red[0].clone()
            # local total = rw * local_residual + (aux terms) / world  ->  global total (for logging only)
            lw = self.pde._loss_weights() if hasattr(self.pde, "_loss_weights") else None
            rw = (lw.get("pde", lw.get("residual", 1.0)) if lw else 1.0)
            losses["total"] = (losses["total"].detach() - rw * local) * world + rw * red[0]

    def _adaptive_total(self, losses):
        """trainer.py:586-684 without the printing: reweight residual/boundary/initial (+ smoothness)."""
        tc = self.config.training
        names = ["residual", "boundary", "initial"]
        if "smoothness" in losses and tc.loss_weights.get("smoothness", 0.0) > 0:
            names.append("smoothness")
        comps = torch.stack([losses[n].detach().reshape(()) for n in names])
        if tc.adaptive_weights.strategy == "lrw":
            norms = []
            for n in names:
                self.optimizer.zero_grad()
                losses[n].backward(retain_graph=True)
                sq = sum(float(p.grad.norm().item()) ** 2 for p in self.model.parameters() if p.grad is not None)
                norms.append(torch.tensor(sq**0.5, device=self.device))
            self.optimizer.zero_grad()
            weights = self.adaptive_weights.update(gradients=torch.stack(norms))
        else:
            weights = self.adaptive_weights.update(losses=comps)
        total = 0
        for i, n in enumerate(names):
            if i < len(weights):
                total = total + weights[i] * losses[n]
        if tc.mode in ("inverse", "data_augmented") and "data" in losses:
            total = total + (tc.loss_weights.get("data", 1.0) or 1.0) * losses["data"]
        w = weights.detach().cpu().numpy()
        if len(w) < 4:
            w = np.concatenate([w, np.zeros(4 - len(w))])
        self.history["loss_weights"].append(w)
        return total

    def train_step(self, x, t):
        """zero_grad -> compute_loss -> backward -> clip -> step  (trainer.py:576-694)."""
        if self._is_lbfgs:
            return self._lbfgs_step(x, t)
        if getattr(self, "_flat", None) is not None:  # parameters live in the flat buffers: same sequence, not captured
            self._manual_launches(x, t)
            return self._manual_losses()
        if self.process_group is not None:
            self._ensure_dp_buffer()
            self._dp_buf.zero_()  # the .grad views stay attached (zero_grad(set_to_none=True) would drop them)
        else:
            self.optimizer.zero_grad()
        losses = self._losses(x, t)
        if self.use_adaptive_weights and self.config.training.mode != "data_only":
            losses["total"] = self._adaptive_total(losses)
        losses["total"].backward()
        self._sync_grads(losses)
        gc = self.config.training.gradient_clipping
        if gc > 0:
            nn.utils.clip_grad_norm_(self.model.parameters(), gc)
        self.optimizer.step()
        return losses

    # ---------------------------------------------------------------- autograd-free step
    def _manual_step_unsupported(self) -> Optional[str]:
        """None when the step can run as the fixed launch sequence below; else the reason (callers fall back)."""
        from ..pdes.pde_base import PDEBase

        tc = self.config.training
        if self._is_lbfgs or getattr(tc, "optimizer", "adam") != "adam":
            return "optimizer is not Adam"
        if self.use_adaptive_weights:
            return "adaptive loss weights"
        if self.rl_agent is not None and not hasattr(self.rl_agent, "action_probabilities"):
            return "RL agent without a device-side action selection"
        if getattr(tc, "collocation_distribution", "uniform") not in ("uniform", "stratified", "residual_based"):
            return "unknown sampler"
        own_loss = type(self.pde).compute_loss is not PDEBase.compute_loss
        if own_loss and type(self.pde)._manual_chain is PDEBase._manual_chain:
            return f"{type(self.pde).__name__} overrides compute_loss without a launch-list form (_manual_chain)"
        if own_loss and (self.pde._loss_weights() or {}).get("smoothness", 0.0) > 0:
            return "smoothness term"
        if self.pde.dimension != 1 or self.pde._has_trainable_coefficients() or self.pde._training_mode() != "forward":
            return "multi-dimensional, inverse or data-driven mode"
        if getattr(self.pde, "observation_data", None):
            return "observation data term"
        ic = getattr(self.pde.config, "initial_condition", None) or {}
        if ic.get("type") == "random":
            return "random initial condition"
        return None

    def _build_flat_state(self):
        """Move the parameters into ONE flat fp32 buffer (each `nn.Parameter` becomes a view of it, same layout as the
        engine's flat gradient) with flat Adam moments beside it; optimizer state of eager steps taken so far comes along."""
        if getattr(self, "_flat", None) is not None:
            return self._flat
        prog = self.model.program()
        offs, n = prog.grad_layout()
        dev = self.device
        theta = torch.zeros(n, dtype=torch.float32, device=dev)
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        steps = 0.0
        with torch.no_grad():
            for p, o in zip(prog.tensors, offs):
                if o < 0:
                    continue
                view = theta[o : o + p.numel()].view_as(p)
                view.copy_(p.data)
                st = self.optimizer.state.get(p, {})
                if "exp_avg" in st:
                    m[o : o + p.numel()].view_as(p).copy_(st["exp_avg"])
                    v[o : o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
                    steps = float(st["step"])
                p.data = view
        g = self.optimizer.param_groups[0]
        self.optimizer._opt_called = True  # the flat Adam kernel steps from now on; schedulers only read / write param_groups
        lw = self.pde._loss_weights()
        rw = lw.get("pde", lw.get("residual", 1.0)) if lw else 1.0
        self._flat = {
            "theta": theta, "m": m, "v": v, "grad": torch.zeros(n + 4, dtype=torch.float32, device=dev),
            "step": torch.tensor([steps], dtype=torch.float32, device=dev),
            "lr": torch.tensor([g["lr"]], dtype=torch.float32, device=dev),
            "scratch": torch.zeros(64, dtype=torch.float32, device=dev), "n": n, "rw": float(rw), "chains": {},
            "grad_side": torch.zeros(n, dtype=torch.float32, device=dev),
            "summary": torch.zeros(4, dtype=torch.float32, device=dev),
            "betas": g["betas"], "eps": g["eps"], "wd": g["weight_decay"],
        }
        return self._flat

    def _chain(self, n_batch: int, world: int = 1):
        """The boundary / initial part of `compute_loss` as a launch-list description (`PDEBase._manual_chain`: fixed
        evaluation points, stream set, loss terms), cached per batch length (HeatEquation sizes its point sets from it
        when the config carries no counts) with its persistent buffers."""
        F = self._flat
        key = (int(n_batch), int(world))
        ch = F["chains"].get(key)
        if ch is None:
            ch = dict(self.pde._manual_chain(int(n_batch)))
            dev = self.device
            K, npts = 1 + ch["nt"] + ch["nx"], ch["x"].shape[0]
            ch["term_losses"] = torch.zeros(len(ch["terms"]), dtype=torch.float32, device=dev)
            ch["cot"] = torch.zeros(K, npts, dtype=torch.float32, device=dev)
            ch["terms_dp"] = [(lo, hi, st, pr, tg, w / world) for lo, hi, st, pr, tg, w in ch["terms"]]
            F["chains"][key] = ch
        return ch

    def _manual_launches(self, x, t, side=None):
        """One optimiser step as a fixed launch sequence, no autograd (pinnrl/training/trainer.py:686-698 with
        pinnrl/pdes/pde_base.py:1086-1235 inlined): zero the flat gradient; residual + mean l(r) + d/dtheta in one
        launch; network values on the 200 boundary + 100 initial points; their loss terms and cotangents; their
        reverse sweep; clip_grad_norm_ + Adam over the flat buffers.

        `side`: a second stream for the boundary / initial chain (forward, loss terms, reverse sweep into its own
        gradient buffer).  The residual launch leaves most CUs idle during its last tile round (1 555 tiles on 256 CUs:
        19 CUs run a seventh tile), which is where the side chain's 10 tiles then run; the two gradients are added
        after the join, so no two launches ever write the same buffer concurrently."""
        F = self._flat
        prog = self.model.program()
        pd = self.pde._pde_desc()
        n, N = F["n"], x.shape[0]
        loss_name, delta = self.pde._loss_function_name(), self.pde._huber_delta()
        if self.process_group is not None:
            return self._manual_launches_dp(x, t, F, prog, pd, n, N, loss_name, delta)

        ch = self._chain(N)

        def boundary_chain(grad, summary):
            u = _E.jets_forward(prog, ch["x"], ch["t"], ch["nt"], ch["nx"])
            _E.jet_losses(u, ch["terms"], loss_name, delta, ch["term_losses"], ch["cot"],
                          residual_sum=F["grad"][n : n + 1] if summary else None, residual_scale=1.0 / float(N),
                          residual_weight=F["rw"], n_boundary_terms=ch["n_bc"], summary4=F["summary"] if summary else None)
            _E.jets_backward(prog, ch["x"], ch["t"], ch["nt"], ch["nx"], ch["cot"], grad)
            return u

        F["grad"].zero_()
        if side is None:
            _E.residual_loss_grad(prog, pd, x, t, F["rw"] / float(N), F["grad"][:n], loss_sum=F["grad"][n : n + 1])
            boundary_chain(F["grad"][:n], True)
        else:
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                F["grad_side"].zero_()
                u = boundary_chain(F["grad_side"], False)
            _E.residual_loss_grad(prog, pd, x, t, F["rw"] / float(N), F["grad"][:n], loss_sum=F["grad"][n : n + 1])
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                u.record_stream(main)
            F["grad"][:n].add_(F["grad_side"])
            # the {residual, boundary, initial, total} summary needs the residual launch's loss sum: the (3 us) loss-term
            # kernel runs once more here; it rewrites the same cotangents
            _E.jet_losses(u, ch["terms"], loss_name, delta, ch["term_losses"], ch["cot"], residual_sum=F["grad"][n : n + 1],
                          residual_scale=1.0 / float(N), residual_weight=F["rw"], n_boundary_terms=ch["n_bc"], summary4=F["summary"])
        _E.adam_clip_step(F["theta"], F["grad"], F["m"], F["v"], F["lr"], F["step"], F["scratch"], beta1=F["betas"][0],
                          beta2=F["betas"][1], eps=F["eps"], weight_decay=F["wd"],
                          max_norm=float(self.config.training.gradient_clipping))

    def _manual_launches_dp(self, x, t, F, prog, pd, n, N, loss_name, delta):
        """The same launch list under a process group (one process per GPU): this rank's contiguous shard of the
        identically-sampled batch goes through the residual launch with the GLOBAL 1/N, the replicated boundary /
        initial chain contributes with weight 1/world, ONE in-place sum all-reduce carries [flat gradient || residual
        loss sum], and every rank applies the same clip + Adam update: replicas stay bit-identical."""
        pg = self.process_group
        world = torch.distributed.get_world_size(pg)
        xs, ts, _ = _D.shard_points(x, t, pg)
        ch = self._chain(N, world)
        F["grad"].zero_()
        _E.residual_loss_grad(prog, pd, xs, ts, F["rw"] / float(N), F["grad"][:n], loss_sum=F["grad"][n : n + 1])
        u = _E.jets_forward(prog, ch["x"], ch["t"], ch["nt"], ch["nx"])
        _E.jet_losses(u, ch["terms_dp"], loss_name, delta, ch["term_losses"], ch["cot"])
        _E.jets_backward(prog, ch["x"], ch["t"], ch["nt"], ch["nx"], ch["cot"], F["grad"][:n])
        torch.distributed.all_reduce(F["grad"], op=torch.distributed.ReduceOp.SUM, group=pg)
        # loss terms with the unscaled weights and the reduced (global) residual sum, for the step's summary
        _E.jet_losses(u, ch["terms"], loss_name, delta, ch["term_losses"], ch["cot"], residual_sum=F["grad"][n : n + 1],
                      residual_scale=1.0 / float(N), residual_weight=F["rw"], n_boundary_terms=ch["n_bc"], summary4=F["summary"])
        _E.adam_clip_step(F["theta"], F["grad"], F["m"], F["v"], F["lr"], F["step"], F["scratch"], beta1=F["betas"][0],
                          beta2=F["betas"][1], eps=F["eps"], weight_decay=F["wd"],
                          max_norm=float(self.config.training.gradient_clipping))

    def _manual_losses(self, static: bool = False):
        """{residual, boundary, initial, total} of the last manual step.  `static=True` hands out views of the persistent
        summary buffer (what a captured graph refreshes in place); otherwise independent copies, so that a caller may
        keep one per step (`train()` averages them per epoch)."""
        s = self._flat["summary"]
        if not static:
            s = s.clone()
        return {"residual": s[0], "boundary": s[1], "initial": s[2], "total": s[3]}

    def get_training_history(self):  # trainer.py:966-972
        return self.history

    def make_graphed_step(self, batch_size: int, warmup: int = 2):
        """Capture ONE whole training step in a HIP graph and return `(replay, losses)`: `replay()` runs a step on a
        fresh device-side sample, `losses` is the dict of STATIC loss tensors it refreshes.

        The captured step contains no autograd at all: it is `_manual_launches` — a handful of kernels of this library (the boundary /
        initial chain forked onto a second stream beside the residual launch) plus the sampler's element-wise ones — writing into persistent flat buffers (parameters, gradient, Adam
        moments; the parameters of the model become views of the flat buffer).  Nothing in it depends on autograd
        nodes of earlier eager steps, so it is safe to call after any number of `train_step`s (round 1's capture of
        `loss.backward()` crashed on a stale AccumulateGrad node).  The learning rate lives in a device scalar that
        `train` refreshes after each scheduler step.  Steps the fixed sequence does not cover (L-BFGS, adaptive
        weights, RL / residual-based sampling, data-parallel, PDEs with their own compute_loss) raise."""
        why = self._manual_step_unsupported()
        if why is None and self.process_group is not None:
            why = "data-parallel training (a collective inside the capture)"
        if why is not None:
            raise NotImplementedError(f"graph capture covers the plain Adam step only ({why})")
        self._build_flat_state()

        overlap = torch.cuda.Stream(device=self.device)  # boundary / initial chain, forked and joined inside the step

        def step():
            x, t = self._sample(batch_size)
            self._manual_launches(x, t, side=overlap)

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):  # sizes the engine's scratch (both streams) and the allocator pools before capture
                step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        self._step_graph = graph  # keeps the captured allocations alive
        return graph.replay, self._manual_losses(static=True)

    def set_learning_rate(self, lr: float) -> None:
        """Propagate a scheduler's learning rate to the device scalar the captured / manual step reads."""
        if getattr(self, "_flat", None) is not None:
            self._flat["lr"].fill_(float(lr))

    # ---------------------------------------------------------------- the loop (trainer.py:391-964)
    def train(self, num_epochs: int, batch_size: int, num_points: int, experiment_dir: str = None):
        self.model.train()
        if self._is_lbfgs and batch_size != num_points:
            batch_size = num_points  # L-BFGS needs full-batch closures (trainer.py:455-461)
        if getattr(self, "_flat", None) is None and self.fast_step is not False:
            why = self._manual_step_unsupported()
            if why is None and torch.device(self.device).type == "cuda":
                self._build_flat_state()  # from here on `train_step` is the fixed launch list (no autograd)
            elif self.fast_step:
                raise NotImplementedError(f"fast_step=True, but the autograd-free step does not cover this configuration ({why})")
        trainable = dict(getattr(self.pde, "_trainable_params", {}))
        for name in trainable:
            self.history.setdefault(f"param_{name}", [])
        start = datetime.now()
        for epoch in range(num_epochs):
            self.model.train()
            step_losses = []
            losses = None
            for _ in range(num_points // batch_size):
                x, t = self._sample(batch_size)
                losses = self.train_step(x, t)
                step_losses.append(losses["total"].detach().clone())  # the manual step's losses may alias one buffer
                if self.log_every_step:
                    self.points_history.append(torch.cat([x, t], dim=1).cpu().numpy())
            avg = float(torch.stack(step_losses).mean().item())  # ZeroDivisionError upstream when there are no steps
            self._update_scheduler(avg)
            lr = self.optimizer.param_groups[0]["lr"]
            self.set_learning_rate(lr)
            row = {"train_loss": avg, "residual_loss": float(losses["residual"]), "boundary_loss": float(losses["boundary"]),
                   "initial_loss": float(losses["initial"]), "learning_rate": lr}
            if "data" in losses:
                self.history.setdefault("data_loss", [])
                row["data_loss"] = float(losses["data"])
            for name, p in trainable.items():
                row[f"param_{name}"] = float(p.detach().cpu().item())
            for k, v in row.items():
                if k in self.history:
                    self.history[k].append(v)
            if experiment_dir and self.viz_frequency and epoch % self.viz_frequency == 0:
                self._save_live_snapshot(experiment_dir, epoch)
            if epoch % self.validation_frequency == 0:
                val = self._compute_validation_loss()
                self.history["val_loss"].append(val["total_loss"])
                if self.early_stopping_enabled:
                    if val["total_loss"] < self.best_val_loss - 1e-6:
                        self.best_val_loss, self.patience_counter = val["total_loss"], 0
                    else:
                        self.patience_counter += 1
                    if self.patience_counter >= self.patience:
                        self.logger.info(f"Early stopping triggered at epoch {epoch + 1}")
                        break
            if (self._optimizer_type == "adam_lbfgs" and not self._is_lbfgs and self._switch_epoch is not None
                    and (epoch + 1) >= self._switch_epoch):
                self._switch_to_lbfgs()
                batch_size = num_points
        self.training_time_minutes = (datetime.now() - start).total_seconds() / 60.0
        return self.history
