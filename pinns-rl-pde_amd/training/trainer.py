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
                 process_group=None, log_every_step: bool = False):
        self.device = device or (config.device if hasattr(config, "device") else torch.device("cpu"))
        self.model = model.to(self.device)
        self.pde = pde
        self.config = config
        self.validation_frequency = validation_frequency
        self.logger = logging.getLogger(__name__)
        self.process_group = process_group
        self.log_every_step = log_every_step
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
        return {"total_loss": losses["total"].item(), "residual_loss": losses["residual"].item(),
                "boundary_loss": losses["boundary"].item(), "initial_loss": losses["initial"].item()}

    # ---------------------------------------------------------------- one step
    def _sample(self, batch_size: int):
        strategy = "adaptive" if self.rl_agent is not None else self.config.training.collocation_distribution
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

    def _sync_grads(self, losses=None):
        """ONE all-reduce of [gradients || residual loss]; the reduced (global) residual replaces the local shard's."""
        if self.process_group is None:
            return
        scalars = [losses["residual"]] if losses is not None else None
        red = _D.all_reduce_gradients(self._collect_optimizable_params(), self.process_group, scalars=scalars)
        if red is not None and losses is not None:
            world = torch.distributed.get_world_size(self.process_group)
            local = losses["residual"].detach()
            losses["residual"] = red[0]
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

    # ---------------------------------------------------------------- graph-captured step
    def make_graphed_step(self, batch_size: int, warmup: int = 3):
        """Capture ONE whole training step — fresh device-side sample, compute_loss, backward, clip_grad_norm_, Adam —
        in a HIP graph and return `(replay, losses)`: `replay()` runs a step, `losses` is the dict of STATIC loss
        tensors it refreshes.  Same arithmetic as `train_step` on `_sample(batch_size)`; what it removes is the
        per-step host work (~60 launch-bound kernels and the Python between them).  Requirements of graph capture:
        Adam (made capturable: its step counter moves to the device), no adaptive loss weights, no RL/RAR sampling
        (their control flow is host-side), gradients kept allocated (`zero_grad(set_to_none=False)`), one process,
        and no autograd graph of an earlier eager step still referenced by the caller (drop old loss tensors first:
        torch's capture of a backward that meets a stale AccumulateGrad node crashes on this ROCm build)."""
        if self._is_lbfgs or self.use_adaptive_weights or self.process_group is not None or self.rl_agent is not None:
            raise NotImplementedError("graph capture covers the single-process Adam step without adaptive loss weights "
                                      "or RL-driven sampling")
        if getattr(self.config.training, "collocation_distribution", "uniform") not in ("uniform", "stratified"):
            raise NotImplementedError("graph capture needs a host-independent sampler (uniform / stratified)")
        for g in self.optimizer.param_groups:
            g["capturable"] = True
        for st in self.optimizer.state.values():  # steps taken eagerly so far: counters move to the device
            if "step" in st and not st["step"].is_cuda:
                st["step"] = st["step"].to(self.device)
        gc = self.config.training.gradient_clipping

        def step():
            x, t = self._sample(batch_size)
            self.optimizer.zero_grad(set_to_none=False)
            losses = self._losses(x, t)
            losses["total"].backward()
            if gc > 0:
                nn.utils.clip_grad_norm_(self.model.parameters(), gc)
            self.optimizer.step()
            return losses

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):  # allocates gradients / optimizer state before capture
                step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            losses = step()
        self._step_graph = graph  # keeps the captured allocations alive
        return graph.replay, {k: v.detach() for k, v in losses.items()}

    # ---------------------------------------------------------------- the loop (trainer.py:391-964)
    def train(self, num_epochs: int, batch_size: int, num_points: int, experiment_dir: str = None):
        self.model.train()
        if self._is_lbfgs and batch_size != num_points:
            batch_size = num_points  # L-BFGS needs full-batch closures (trainer.py:455-461)
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
                step_losses.append(losses["total"].detach())
                if self.log_every_step:
                    self.points_history.append(torch.cat([x, t], dim=1).cpu().numpy())
            avg = float(torch.stack(step_losses).mean().item())  # ZeroDivisionError upstream when there are no steps
            self._update_scheduler(avg)
            lr = self.optimizer.param_groups[0]["lr"]
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
