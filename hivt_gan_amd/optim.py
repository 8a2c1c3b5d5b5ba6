"""AdamW for GPU parameters with a short host path.

``torch.optim.AdamW(fused=True)`` rebuilds its tensor lists from ``param_groups`` and ``state`` on every
call and advances one step counter per parameter (several hundred microseconds of host time and a
dozen small launches for a few hundred parameters).  ``FusedAdamW`` is the same optimizer (same state
layout, ``state_dict``, schedulers, ``found_inf`` / ``grad_scale`` attributes) that keeps those lists
between calls and lets the parameters of a group share ONE step counter while their counts agree.
Whenever the set of parameters holding a gradient differs from the cached one, and whenever the state is
exported or replaced, it returns to the stock layout and ``step``.  The arithmetic of a cached step runs
in the library's own multi-tensor kernel (``hivt_adamw_step``; same formula, results equal to the
framework's to rounding); ``use_library_kernel = False`` keeps the framework's kernel."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_DATA_PTR = torch.Tensor.data_ptr


class FusedAdamW(torch.optim.AdamW):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2) -> None:
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        # per group: (params with a gradient, the rest, exp_avgs, exp_avg_sqs, step list, shared step or None)
        self._lists = None
        self.use_library_kernel = True
        self._library_meta = {}  # id(list of params) -> host arrays of state addresses and sizes
        self.library_steps = 0   # group updates that went through the library's kernel
        # per group: a device scalar holding the learning rate the library's kernel reads, and the value it holds.
        # A schedule changes `group["lr"]`; `sync_lr()` carries the change to the device, so a recorded graph of the
        # update keeps replaying across learning-rate changes (the rate is not a launch argument).
        self._lr_dev = {}
        self.last_step_on_library = False  # every group of the last step() went through the library's kernel

    # -- cache management -------------------------------------------------------------------------
    def _untie(self) -> None:
        """Back to one step tensor per parameter (the stock layout) and no cached lists."""
        if self._lists is not None:
            for params, _, _, _, _, shared in self._lists:
                if shared is not None:
                    for p in params:
                        self.state[p]["step"] = shared.clone()
        self._lists = None

    def _cache(self) -> None:
        self._library_meta = {}
        lists = []
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            rest = [p for p in group["params"] if p.grad is None]
            states = [self.state[p] for p in params]
            if any(not st for st in states) or group["amsgrad"] or group["maximize"]:
                self._lists = None
                return
            steps = [st["step"] for st in states]
            shared = None
            if steps and all(s.is_cuda and s.dim() == 0 for s in steps):
                first = float(steps[0])  # one read-back, when the cache is (re)built only
                if all(float(s) == first for s in steps[1:8]) and bool(torch.stack(steps).eq(first).all()):
                    shared = steps[0].clone()
                    for st in states:
                        st["step"] = shared
                    steps = [shared] * len(states)
            lists.append((params, rest, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states], steps,
                          shared))
        self._lists = lists

    def state_dict(self):
        self._untie()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        self._lists = None
        super().load_state_dict(state_dict)
        for st in self.state.values():  # a loaded file may alias the counters of a tied group
            if "step" in st:
                st["step"] = st["step"].clone()

    def add_param_group(self, param_group) -> None:
        if getattr(self, "_lists", None) is not None:
            self._untie()
        super().add_param_group(param_group)

    def zero_grad(self, set_to_none: bool = True) -> None:
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def sync_lr(self) -> None:
        """Writes every group's current learning rate into its device scalar if it has changed.  Called by ``step``
        (outside stream capture) and by whoever replays a recorded step before a replay."""
        for i, group in enumerate(self.param_groups):
            slot = self._lr_dev.get(i)
            if slot is not None and slot[1] != group["lr"] and isinstance(group["lr"], float):
                slot[0].fill_(group["lr"])
                slot[1] = group["lr"]

    def _lr_scalar(self, index: int, group, device) -> torch.Tensor:
        slot = self._lr_dev.get(index)
        if slot is None or slot[0].device != device:
            slot = self._lr_dev[index] = [torch.full((), group["lr"], dtype=torch.float32, device=device), group["lr"]]
        return slot[0]

    # -- the update ---------------------------------------------------------------------------------
    def _library_step(self, group, params, grads, exp_avgs, exp_avg_sqs, step, found_inf, index=0) -> bool:
        """The update of one group through ``hivt_adamw_update`` (up to 96 tensors per launch, 4096-element
        pieces) instead of the framework's multi-tensor kernel, whose 65 536-element pieces leave a few
        hundred small tensors with a handful of busy workgroups per launch.  ``step`` holds the number of updates
        done so far; the call uses ``step + 1`` and advances the counter unless ``found_inf`` is set.  False: not
        applicable here (the caller then runs the framework's kernel)."""
        if not self.use_library_kernel or not isinstance(group["lr"], float) or step.dtype != torch.float32:
            return False
        for g in grads:
            if not g.is_contiguous() or g.dtype != torch.float32:
                return False
        if found_inf is not None and (found_inf.dtype != torch.float32 or not found_inf.is_cuda):
            return False
        count = len(params)
        meta = self._library_meta.get(id(params))
        if meta is None or meta[0] is not params:
            for t in params + exp_avgs + exp_avg_sqs:
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    return False
            array = ctypes.c_uint64 * count
            meta = (params, array, array(*[t.data_ptr() for t in exp_avgs]), array(*[t.data_ptr() for t in exp_avg_sqs]),
                    (ctypes.c_int64 * count)(*[p.numel() for p in params]))
            self._library_meta[id(params)] = meta
        _, array, m_ptrs, v_ptrs, numels = meta
        beta1, beta2 = group["betas"]
        # parameter addresses are read every time: a parameter's storage may have been replaced since
        lr_dev = self._lr_scalar(index, group, params[0].device)
        _lib.call("hivt_adamw_update", array(*map(_DATA_PTR, params)), array(*map(_DATA_PTR, grads)), m_ptrs, v_ptrs,
                  numels, count, step.data_ptr(), found_inf.data_ptr() if found_inf is not None else None,
                  lr_dev.data_ptr(), group["lr"], beta1, beta2, group["eps"], group["weight_decay"],
                  torch.cuda.current_stream(params[0].device).cuda_stream)
        self.library_steps += 1
        return True

    def _init_missing_state(self) -> bool:
        """State for the parameters that hold a gradient and have none yet, in the stock layout (a 0-d float32 step
        counter on the device, two zero moments), so that the first update already takes the cached path.  False if
        a group is configured in a way the cached path does not serve."""
        for group in self.param_groups:
            if group["amsgrad"] or group["maximize"] or group.get("capturable") or group.get("differentiable"):
                return False
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or not p.is_cuda:
                    return False
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is None and self._lists is None and not torch.cuda.is_current_stream_capturing() \
                and self._init_missing_state():
            self._cache()
        lists = self._lists
        if closure is None and lists is not None and len(lists) == len(self.param_groups):
            grads_per_group = []
            for params, rest, _, _, _, _ in lists:
                grads = [p.grad for p in params]
                ok = True
                for g in grads:
                    if g is None:
                        ok = False
                        break
                if ok:
                    for p in rest:
                        if p.grad is not None:
                            ok = False
                            break
                if not ok:
                    grads_per_group = None
                    break
                grads_per_group.append(grads)
            if grads_per_group is not None:
                found_inf = getattr(self, "found_inf", None)
                grad_scale = getattr(self, "grad_scale", None)
                if not torch.cuda.is_current_stream_capturing():
                    self.sync_lr()
                self.last_step_on_library = True
                for index, (group, (params, _, exp_avgs, exp_avg_sqs, steps, shared), grads) in enumerate(
                        zip(self.param_groups, lists, grads_per_group)):
                    if not params:
                        continue
                    beta1, beta2 = group["betas"]
                    # the library's kernel counts the step itself (one launch less on either side of the update)
                    if shared is not None and grad_scale is None and self._library_step(
                            group, params, grads, exp_avgs, exp_avg_sqs, shared, found_inf, index):
                        continue
                    self.last_step_on_library = False
                    if shared is not None:
                        shared.add_(1)
                    else:
                        torch._foreach_add_(steps, 1)
                    torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False,
                                        lr=group["lr"], beta1=beta1, beta2=beta2,
                                        weight_decay=group["weight_decay"], eps=group["eps"], maximize=False,
                                        grad_scale=grad_scale, found_inf=found_inf)
                    if found_inf is not None:  # a skipped update does not count as a step
                        if shared is not None:
                            shared.sub_(found_inf.to(shared.dtype).view(()))
                        else:
                            torch._foreach_sub_(steps, [found_inf] * len(steps))
                return None
        self._untie()
        self.last_step_on_library = False
        out = super().step(closure)
        self._cache()
        return out
