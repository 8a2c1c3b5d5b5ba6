"""HiVT generator + multi-scale critics and the G + D + R1 training step, without Lightning.

``HiVT`` keeps the constructor arguments, ``forward`` contract and state_dict keys of
``models/hivt.py:31-169``.  ``training_step`` restates Case B of the reference
(``models/hivt.py:287-342``): generator forward, supervised losses, critic update with hinge + R1,
generator update with the adversarial term.  Differences, all deliberate:

* ``use_gan`` is honoured.  The reference reads it from ``**kwargs`` where it can never be
  (``models/hivt.py:52`` vs ``:76``), which silently disables the GAN branch it ships.
* the empty-future branch of the supervised losses yields three values (the reference returns two
  at ``models/hivt.py:178-185`` but unpacks three at ``:298``).
* agents without any valid future step are masked out with weights instead of boolean indexing
  (``models/hivt.py:227-231``), which gives the same losses without a data-dependent shape.
* gradients of one backward are reduced across data-parallel ranks by a caller-supplied hook on a
  flat buffer (the reference relies on Lightning's DDP wrapper, ``train.py:69-78``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .critics import LongScaleCritic, MidScaleCritic, ShortScaleCritic
from .decoder import MLPDecoder
from .global_interactor import GlobalInteractor
from .local_encoder import LocalEncoder
from .metrics import ValidationMetrics
from .optim import FusedAdamW
from .critics import multi_supported
from .losses import (AdversarialDiscriminatorLoss, AdversarialGeneratorLoss, LaplaceNLLLoss,
                     SoftTargetCrossEntropyLoss, fused_supervised_losses, join_critic_streams)


class HiVT(nn.Module):

    def __init__(self, historical_steps: int = 20, future_steps: int = 30, num_modes: int = 6, rotate: bool = True,
                 node_dim: int = 2, edge_dim: int = 2, embed_dim: int = 64, num_heads: int = 8, dropout: float = 0.1,
                 num_temporal_layers: int = 4, num_global_layers: int = 3, local_radius: float = 50,
                 parallel: bool = False, lr: float = 5e-4, weight_decay: float = 1e-4, T_max: int = 64,
                 use_gan: bool = False, lambda_adv: float = 0.1, lambda_r1: float = 1.0, critic_steps: int = 1,
                 short_horizon: int = 10, mid_horizon: int = 20, critic_lr: float = 1e-4, **kwargs) -> None:
        super().__init__()
        self.historical_steps = historical_steps
        self.future_steps = future_steps
        self.num_modes = num_modes
        self.rotate = rotate
        self.parallel = parallel
        self.lr, self.weight_decay, self.T_max = lr, weight_decay, T_max
        self.use_gan = bool(use_gan)
        self.lambda_adv, self.lambda_r1 = lambda_adv, lambda_r1
        self.critic_steps = critic_steps
        self.short_horizon, self.mid_horizon = short_horizon, mid_horizon
        self.critic_lr = critic_lr
        self.check_finite = bool(kwargs.get("check_finite", True))
        # sync_free: the guards of the training step stay on the device (no host read-back, no
        # data-dependent control flow), so the host can keep launching ahead of the GPU
        self.sync_free = bool(kwargs.get("sync_free", False))
        # global-norm clipping of the supervised (non-GAN) step; the reference sets 0.5 on its Trainer
        # (train.py:76), where it reaches only that step: the GAN step optimises manually and is not clipped
        clip = kwargs.get("gradient_clip_val", 0.5)
        self.gradient_clip_val = float(clip) if clip else 0.0

        self.local_encoder = LocalEncoder(historical_steps=historical_steps, node_dim=node_dim, edge_dim=edge_dim,
                                          embed_dim=embed_dim, num_heads=num_heads, dropout=dropout,
                                          num_temporal_layers=num_temporal_layers, local_radius=local_radius,
                                          parallel=parallel)
        self.global_interactor = GlobalInteractor(historical_steps=historical_steps, embed_dim=embed_dim,
                                                  edge_dim=edge_dim, num_modes=num_modes, num_heads=num_heads,
                                                  num_layers=num_global_layers, dropout=dropout, rotate=rotate)
        self.decoder = MLPDecoder(local_channels=embed_dim, global_channels=embed_dim, future_steps=future_steps,
                                  num_modes=num_modes, uncertain=True)
        self.reg_loss = LaplaceNLLLoss(reduction="mean")
        self.cls_loss = SoftTargetCrossEntropyLoss(reduction="mean")
        if self.use_gan:
            self.D_short = ShortScaleCritic(horizon=short_horizon, input_dim=2, hidden_dim=128)
            self.D_mid = MidScaleCritic(horizon=mid_horizon, input_dim=2, hidden_dim=256)
            self.D_long = LongScaleCritic(horizon=future_steps, input_dim=2, hidden_dim=256)
            self.d_loss_fn = AdversarialDiscriminatorLoss(lambda_r1=lambda_r1)
            self.g_loss_fn = AdversarialGeneratorLoss(lambda_adv=lambda_adv)

    # ------------------------------------------------------------------------------------------
    @property
    def critics(self) -> Dict[str, nn.Module]:
        return {"short": self.D_short, "mid": self.D_mid, "long": self.D_long}

    def pack_parameters(self) -> int:
        """Lays the parameters of layers that are evaluated as one product (K | V projections, the projections
        of an attention block's centre rows) back to back in memory, so the step hands them to the kernels
        without concatenating them first.  Values, names and ``state_dict`` are unchanged.  Call it with the
        model on its final device and before building optimizers; returns the number of groups packed."""
        groups = 0
        for module in self.modules():
            if hasattr(module, "paired_layers"):
                for layers in module.paired_layers():
                    ops.pack_rows([layer.weight for layer in layers])
                    ops.pack_rows([layer.bias for layer in layers])
                    groups += 1
        return groups

    def _stack_embedding_pairs(self) -> list:
        """The summed branch layers of the three edge embeddings (``models/embedding.py:49-52``) run as one product
        each over the weights side by side: lays those out for this forward pass, all modules in one launch."""
        if not next(self.parameters()).is_cuda:
            return []
        from .embedding import MultipleInputEmbedding
        modules = [m for m in self.modules() if isinstance(m, MultipleInputEmbedding) and m.summed_pair() is not None
                   and m.summed_pair()[0].dtype == torch.float32]
        for m, stacked in zip(modules, ops.stack_pairs([m.summed_pair() for m in modules])):
            m.__dict__["_stacked_pair"] = stacked
        return modules

    def generator_parameters(self) -> List[nn.Parameter]:
        """Parameter order of the generator optimizer (``models/hivt.py:439-442``)."""
        return (list(self.local_encoder.parameters()) + list(self.global_interactor.parameters())
                + list(self.decoder.parameters()))

    def critic_parameters(self) -> List[nn.Parameter]:
        """Parameter order of the critic optimizer (``models/hivt.py:448-451``)."""
        return list(self.D_short.parameters()) + list(self.D_mid.parameters()) + list(self.D_long.parameters())

    def configure_optimizers(self):
        """AdamW (all generator parameters decayed, ``models/hivt.py:428,444``) with cosine annealing;
        a second AdamW without decay for the critics (``models/hivt.py:453``)."""
        fused = next(self.parameters()).is_cuda

        def adamw(params, lr, weight_decay):
            if fused:
                return FusedAdamW(params, lr=lr, weight_decay=weight_decay)
            return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)

        if not self.use_gan:
            opt = adamw(self.parameters(), self.lr, self.weight_decay)
            return [opt], [torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=self.T_max, eta_min=0.0)]
        opt_g = adamw(self.generator_parameters(), self.lr, self.weight_decay)
        opt_d = adamw(self.critic_parameters(), self.critic_lr, 0.0)
        return [opt_g, opt_d], [torch.optim.lr_scheduler.CosineAnnealingLR(opt_g, T_max=self.T_max, eta_min=0.0)]

    # ------------------------------------------------------------------------------------------
    def forward(self, data) -> Tuple[torch.Tensor, torch.Tensor]:
        """(y_hat [F, N, H, 4], pi [N, F]).  Like the reference, rotates ``data.y`` into each agent's
        frame IN PLACE on the batch object and stores ``rotate_mat`` on it (``models/hivt.py:151-169``):
        re-using one batch object for two forwards needs ``data.y`` restored in between."""
        data["_prep"] = None
        if self.rotate and ops.batch_prepare_supported(data, self.historical_steps) \
                and int(data["padding_mask"].shape[1]) == self.historical_steps + self.future_steps:
            # rotation matrices, rotated targets and histories, lane codes and the loss weights in one launch
            prep = ops.batch_prepare(data, self.historical_steps)
            if prep["y"] is not None:
                data["y"] = prep["y"]
            data["rotate_mat"] = prep["rotate_mat"]
            data["_prep"] = prep
        elif self.rotate:
            ang = data["rotate_angles"]
            sin, cos = torch.sin(ang), torch.cos(ang)
            rotate_mat = torch.stack([cos, -sin, sin, cos], dim=-1).view(-1, 2, 2)
            if data.get("y") is not None:
                # [N, H, 2] x [N, 2, 2] written out: a batch of N 2x2 products is not worth a GEMM launch
                y = data["y"]
                data["y"] = y[..., 0:1] * rotate_mat[:, None, 0, :] + y[..., 1:2] * rotate_mat[:, None, 1, :]
            data["rotate_mat"] = rotate_mat
        else:
            data["rotate_mat"] = None
        stacked_for = self._stack_embedding_pairs()
        try:
            local_embed = self.local_encoder(data=data)
            global_embed = self.global_interactor(data=data, local_embed=local_embed)
        finally:
            for m in stacked_for:  # (a module that did not run keeps nothing for a later pass with other weights)
                m.__dict__.pop("_stacked_pair", None)
        return self.decoder(local_embed=local_embed, global_embed=global_embed)

    def _supervised_losses(self, data, y_hat: torch.Tensor, pi: torch.Tensor):
        """(reg_loss, cls_loss, best_mode), ``models/hivt.py:174-209`` on the fused kernel."""
        return fused_supervised_losses(y_hat, pi, data["y"], data["padding_mask"], self.historical_steps)

    def _real_fake(self, data, y_hat: torch.Tensor, best_mode: torch.Tensor):
        """Real / best-mode fake futures of every agent plus the keep mask of agents with a valid
        future (``models/hivt.py:215-231``; horizons are cut inside the critics)."""
        n = y_hat.shape[1]
        fake = y_hat[best_mode, torch.arange(n, device=y_hat.device), :, :2]
        keep = (~data["padding_mask"][:, self.historical_steps:]).any(dim=1)
        return data["y"], fake, keep

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def validation_step(self, data, metrics: Optional[ValidationMetrics] = None) -> Dict[str, torch.Tensor]:
        """Forward pass plus the reference's validation quantities (``models/hivt.py:347-393``): the
        regression loss of the best mode over all valid agents, and for each scene's target agent
        (``agent_index``) the trajectory of the mode with the smallest final displacement, from which
        minADE / minFDE / minMR and the realism metrics are accumulated into ``metrics``.  Nothing is
        read back: a batch without any valid future step enters the running loss with weight 0 (the
        reference returns early there).  Returns the batch's own values as device tensors."""
        y_hat, pi = self(data)
        reg_loss, _cls, _best = self._supervised_losses(data, y_hat, pi)
        has_future = (~data["padding_mask"][:, self.historical_steps:]).any()
        usable = has_future & torch.isfinite(reg_loss)
        agent = data["agent_index"]
        y_hat_agent = y_hat[:, agent, :, :2]                                     # [F, B, H, 2]
        y_agent = data["y"][agent]                                               # [B, H, 2]
        fde = torch.norm(y_hat_agent[:, :, -1] - y_agent[:, -1], p=2, dim=-1)    # [F, B]
        best_mode_agent = fde.argmin(dim=0)
        best = y_hat_agent[best_mode_agent, torch.arange(agent.shape[0], device=y_hat.device)]
        if metrics is not None:
            metrics.val_reg_loss._add(torch.where(usable, reg_loss, torch.zeros_like(reg_loss)), usable)
            metrics.minADE.update(best, y_agent)
            metrics.minFDE.update(best, y_agent)
            metrics.minMR.update(best, y_agent)
            metrics.jerk.update(best)
            metrics.speed_violation.update(best)
            metrics.endpoint_diversity.update(y_hat_agent)
        ade = torch.norm(best - y_agent, p=2, dim=-1).mean(dim=-1)
        return {"val_reg_loss": reg_loss, "val_minADE": ade.mean(), "val_minFDE": fde.min(dim=0).values.mean(),
                "best_mode_agent": best_mode_agent}

    # ------------------------------------------------------------------------------------------
    def training_step(self, data, opt_g: torch.optim.Optimizer, opt_d: Optional[torch.optim.Optimizer] = None,
                      sync_grads: Optional[Callable[..., None]] = None) -> Dict[str, torch.Tensor]:
        """One optimisation step; returns detached loss tensors (no host read-back of losses).

        ``sync_grads(group, bad)`` is called after each backward with ``"critic"`` or ``"generator"`` and
        must average that group's gradients across data-parallel ranks; ``bad`` is ``None`` or a device
        bool telling that this rank's gradients must enter the average as zeros.  A rank whose batch
        trips a guard still calls it so every rank enters the same collectives
        (``models/hivt.py:262-265,301-310``).

        Guards: a NaN target or a non-finite prediction discards the batch.  By default they are
        evaluated on the host (two read-backs per step).  With ``sync_free`` they stay on the device:
        without data parallelism the optimizers skip their update through ``found_inf`` (fused AdamW),
        with it the rank contributes zero gradients -- the same outcomes without any synchronisation."""
        with ops.seed_scope():  # one draw of the host generator for all dropout seeds of the step
            if self.sync_free:
                return self._training_step_sync_free(data, opt_g, opt_d, sync_grads)
            return self._training_step_guarded(data, opt_g, opt_d, sync_grads)

    def _training_step_guarded(self, data, opt_g, opt_d, sync_grads) -> Dict[str, torch.Tensor]:

        def skip(reason: str) -> Dict[str, torch.Tensor]:
            # zero-gradient participation keeps data-parallel ranks in lock-step: the same sequence of
            # collectives as a healthy rank -- critic_steps critic reductions, then the generator's
            schedule = [("critic", opt_d)] * (self.critic_steps if self.use_gan else 0) + [("generator", opt_g)]
            for group, opt in schedule:
                if opt is None:
                    continue
                if sync_grads is not None:
                    # the other ranks step with the averaged gradient; so must this one, or the
                    # replicas drift apart.  Without data parallelism a skipped batch changes nothing.
                    opt.zero_grad(set_to_none=False)
                    sync_grads(group, None)
                    opt.step()
            return {"skipped": torch.ones((), device=data["y"].device)}

        if self.check_finite and bool(torch.isnan(data["y"]).any()):  # models/hivt.py:264
            return skip("nan target")
        y_hat, pi = self(data)
        if self.check_finite and not bool(torch.isfinite(y_hat).all()):  # models/hivt.py:272,294
            return skip("non-finite prediction")
        return self._losses_and_updates(data, y_hat, pi, opt_g, opt_d, sync_grads, None)

    def _training_step_sync_free(self, data, opt_g, opt_d, sync_grads) -> Dict[str, torch.Tensor]:
        y_raw = data["y"]
        y_hat, pi = self(data)
        plan = data.get("edge_plan")
        # a fixed-capacity edge list that was truncated discards the batch like a tripped guard
        overflow = plan.overflow if (plan is not None and plan.overflow is not None) else None
        if y_hat.is_cuda:
            bad = ops.step_guards(y_raw if self.check_finite else None, y_hat if self.check_finite else None, overflow) \
                if (self.check_finite or overflow is not None) else torch.zeros((), dtype=torch.bool, device=y_hat.device)
        else:
            bad = torch.zeros((), dtype=torch.bool, device=y_hat.device)
            if self.check_finite:
                bad = torch.isnan(y_raw).any() | ~torch.isfinite(y_hat).all()
            if overflow is not None:
                bad = bad | overflow
        logs = self._losses_and_updates(data, y_hat, pi, opt_g, opt_d, sync_grads, bad)
        if "skipped" not in logs:
            logs["skipped"] = bad.float()
        return logs

    def _losses_and_updates(self, data, y_hat, pi, opt_g, opt_d, sync_grads, bad) -> Dict[str, torch.Tensor]:
        found_inf = bad.float() if bad is not None else None

        def backward(loss: torch.Tensor) -> None:
            if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
                loss.backward(ops.unit_scalar(loss))  # (a cached one: no fill per backward pass)
            else:
                loss.backward()

        def arm(group: str) -> None:
            # a data-parallel reducer that overlaps its collectives with the backward pass wants to know
            # before the pass starts
            if sync_grads is not None and hasattr(sync_grads, "begin"):
                sync_grads.begin(group, bad)

        def update(group: str, opt: torch.optim.Optimizer, clip: bool = False) -> None:
            if sync_grads is not None:
                share = sync_grads(group, bad)
                if bad is not None and torch.is_tensor(share):
                    # every rank discarded its batch: skip the update as the single-process step does
                    opt.found_inf = (share == 0).to(torch.float32)
            elif found_inf is not None:
                opt.found_inf = found_inf  # read by the fused AdamW kernels: 1 leaves parameters and state untouched
            if clip and self.gradient_clip_val:
                # the supervised step is clipped by global norm, after the data-parallel average and before
                # the update, as the reference's Trainer(gradient_clip_val=0.5) does for automatic
                # optimisation (train.py:76); device-side, no read-back
                torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"] if p.grad is not None],
                                               self.gradient_clip_val, foreach=True)
            opt.step()

        packed = None
        if y_hat.is_cuda and y_hat.dtype == torch.float32 and y_hat.shape[-1] == 4:
            # the two supervised losses as one tensor, and (GAN step) the best mode's locations out of the same node
            packed, best_mode, fake = ops.supervised_loss_packed(y_hat, pi, data["y"], data["padding_mask"],
                                                                 self.historical_steps, want_best_traj=self.use_gan)
            plain = packed.detach()
            reg_loss, cls_loss = plain[0], plain[1]
            logs: Dict[str, torch.Tensor] = {"reg_loss": reg_loss, "cls_loss": cls_loss}
        else:
            reg_loss, cls_loss, best_mode = self._supervised_losses(data, y_hat, pi)
            logs = {"reg_loss": reg_loss.detach(), "cls_loss": cls_loss.detach()}

        if not self.use_gan:  # Case A, models/hivt.py:270-285
            loss = packed[0] + packed[1] if packed is not None else reg_loss + cls_loss
            opt_g.zero_grad(set_to_none=True)
            arm("generator")
            backward(loss)
            update("generator", opt_g, clip=True)
            logs["loss"] = loss.detach()
            return logs

        prep = data.get("_prep")
        if packed is not None:
            real = data["y"]
            fused_losses = len(self.critics) <= 4 and multi_supported(self.critics.values(), real, fake)
            if prep is not None and fused_losses:
                # (the fused objectives count the kept rows themselves)
                keep, w, cnt = None, prep["keep_w"], None
            elif prep is not None:
                keep, w = None, prep["keep_w"]
                cnt = w.sum().clamp(min=1.0)
            else:
                keep = (~data["padding_mask"][:, self.historical_steps:]).any(dim=1)
                w = keep.to(real.dtype)
                cnt = w.sum().clamp(min=1.0)
        else:
            real, fake, keep = self._real_fake(data, y_hat, best_mode)
            w = cnt = None
        critic_params = self.critic_parameters()
        for _ in range(self.critic_steps):  # models/hivt.py:305-314
            d_loss, d_logs = self.d_loss_fn.forward_masked(self.critics, real, fake, keep, w, cnt)
            opt_d.zero_grad(set_to_none=True)
            arm("critic")
            backward(d_loss)
            if real.is_cuda:
                join_critic_streams(real.device)
            update("critic", opt_d)
        # generator update (models/hivt.py:317-327); critic weight gradients of this backward would be
        # discarded by the next opt_d.zero_grad(), so they are not computed at all
        for p in critic_params:
            p.requires_grad_(False)
        try:
            if packed is not None and len(self.critics) <= 4 and multi_supported(self.critics.values(), fake):
                g_total, g_adv, g_logs = self.g_loss_fn.total_masked(self.critics, fake, packed, w, cnt)
            else:
                g_adv, g_logs = self.g_loss_fn.forward_masked(self.critics, fake, keep, w, cnt)
                g_total = (packed[0] + packed[1] if packed is not None else reg_loss + cls_loss) + g_adv
            opt_g.zero_grad(set_to_none=True)
            arm("generator")
            backward(g_total)
            if fake.is_cuda:
                join_critic_streams(fake.device)
        finally:
            for p in critic_params:
                p.requires_grad_(True)
        update("generator", opt_g)
        logs.update({"d_loss": d_loss.detach(), "g_adv": g_adv.detach(), "g_total": g_total.detach()})
        if found_inf is not None:
            logs["skipped"] = found_inf
        logs.update(d_logs)
        logs.update(g_logs)
        return logs
