"""AdamW for GPU parameters with a short host path.

``torch.optim.AdamW(fused=True)`` rebuilds its tensor lists from ``param_groups`` and ``state`` on every
call (several hundred microseconds of host time for a few hundred parameters).  ``FusedAdamW`` is the
same optimizer (same state layout, ``state_dict``, schedulers, ``found_inf`` / ``grad_scale``
attributes, same multi-tensor kernel) that keeps those lists between calls; whenever the set of
parameters holding a gradient differs from the cached one it defers to the stock ``step``."""
from __future__ import annotations

import torch


class FusedAdamW(torch.optim.AdamW):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2) -> None:
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._lists = None  # per group: (params, exp_avgs, exp_avg_sqs, steps)

    def _cache(self) -> None:
        lists = []
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            states = [self.state[p] for p in params]
            if any(not st for st in states) or group["amsgrad"] or group["maximize"]:
                self._lists = None
                return
            lists.append((params, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states],
                          [st["step"] for st in states]))
        self._lists = lists

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._lists = None

    def add_param_group(self, param_group) -> None:
        super().add_param_group(param_group)
        self._lists = None

    @torch.no_grad()
    def step(self, closure=None):
        lists = self._lists
        if closure is None and lists is not None and len(lists) == len(self.param_groups):
            grads_per_group = []
            for group, (params, _, _, _) in zip(self.param_groups, lists):
                grads = [p.grad for p in params]
                with_grad = sum(p.grad is not None for p in group["params"])
                if with_grad != len(params) or any(g is None for g in grads):
                    grads_per_group = None
                    break
                grads_per_group.append(grads)
            if grads_per_group is not None:
                found_inf = getattr(self, "found_inf", None)
                grad_scale = getattr(self, "grad_scale", None)
                for group, (params, exp_avgs, exp_avg_sqs, steps), grads in zip(self.param_groups, lists,
                                                                              grads_per_group):
                    if not params:
                        continue
                    beta1, beta2 = group["betas"]
                    torch._foreach_add_(steps, 1)
                    torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False,
                                        lr=group["lr"], beta1=beta1, beta2=beta2, weight_decay=group["weight_decay"],
                                        eps=group["eps"], maximize=False, grad_scale=grad_scale, found_inf=found_inf)
                    if found_inf is not None:  # a skipped update does not count as a step
                        torch._foreach_sub_(steps, [found_inf] * len(steps))
                return None
        out = super().step(closure)
        self._cache()
        return out
