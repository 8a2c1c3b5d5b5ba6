"""Multi-scale trajectory critics: spectral-normalised MLPs over flattened [B, T*2] trajectories.

State_dict layout and behaviour follow ``models/critics/base_critic.py:13-44`` (hook-based
``torch.nn.utils.spectral_norm``: one power iteration per training-mode forward; keys
``mlp.{0,3,6,9}.{weight_orig,weight_u,weight_v,bias}``, ``out.*``) and the three horizon
wrappers (``short_scale_critic.py``, ``mid_scale_critic.py``, ``long_scale_critic.py``).

The spectral normalisation itself is restated in ``SpectralLinear`` instead of being installed as a
hook: on the GPU all layers of a critic are normalised by ONE kernel launch per forward call (the hook
issues about a dozen tiny launches per layer and call, nine calls per training step); for CPU tensors
the same arithmetic runs on framework ops.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_EPS = 1e-12  # default of torch.nn.utils.spectral_norm


class SpectralLinear(nn.Module):
    """``spectral_norm(nn.Linear(in_features, out_features))`` with the hook's parameter / buffer names
    and initialisation (``weight_orig``, ``bias``, ``weight_u``, ``weight_v``)."""

    def __init__(self, in_features: int, out_features: int) -> None:
        super().__init__()
        lin = nn.Linear(in_features, out_features)  # same initial weights and RNG use as the reference
        self.in_features, self.out_features = in_features, out_features
        self.bias = lin.bias
        self.weight_orig = nn.Parameter(lin.weight.data)
        with torch.no_grad():
            u = F.normalize(lin.weight.new_empty(out_features).normal_(0, 1), dim=0, eps=_EPS)
            v = F.normalize(lin.weight.new_empty(in_features).normal_(0, 1), dim=0, eps=_EPS)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)

    def normalized_weight(self) -> torch.Tensor:
        """Framework-op form of the hook's ``compute_weight`` (one power iteration in training mode)."""
        w, u, v = self.weight_orig, self.weight_u, self.weight_v
        if self.training:
            with torch.no_grad():
                v = F.normalize(torch.mv(w.t(), u), dim=0, eps=_EPS, out=v)
                u = F.normalize(torch.mv(w, v), dim=0, eps=_EPS, out=u)
                u, v = u.clone(memory_format=torch.contiguous_format), v.clone(memory_format=torch.contiguous_format)
        sigma = torch.dot(u, torch.mv(w, v))
        return w / sigma

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            (w,) = ops.spectral_normalize([self.weight_orig], [self.weight_u], [self.weight_v], self.training, _EPS)
        else:
            w = self.normalized_weight()
        return F.linear(x, w, self.bias)


class BaseTrajectoryCritic(nn.Module):

    def __init__(self, input_dim: int = 2, hidden_dim: int = 128, num_layers: int = 3, dropout: float = 0.1) -> None:
        super().__init__()
        blocks, width = [], input_dim
        for _ in range(num_layers):
            blocks += [SpectralLinear(width, hidden_dim), nn.LeakyReLU(0.2, inplace=True), nn.Dropout(dropout)]
            width = hidden_dim
        self.mlp = nn.Sequential(*blocks)
        self.out = SpectralLinear(hidden_dim, 1)

    def _spectral_layers(self) -> List[SpectralLinear]:
        return [m for m in self.mlp if isinstance(m, SpectralLinear)] + [self.out]

    def _activation(self):
        slope, drop = 0.01, 0.0
        for mod in self.mlp:
            if isinstance(mod, nn.LeakyReLU):
                slope = mod.negative_slope
            elif isinstance(mod, nn.Dropout):
                drop = mod.p
        return slope, drop

    def forward(self, traj: torch.Tensor) -> torch.Tensor:
        """traj [B, T, D] -> realism score [B]."""
        if traj.is_cuda:
            return self._scores_gpu(traj, False)
        h = traj.reshape(traj.shape[0], -1)
        layers = self._spectral_layers()
        weights = [m.normalized_weight() for m in layers]
        it = iter(zip(layers, weights))
        for mod in self.mlp:
            if isinstance(mod, SpectralLinear):
                lin, w = next(it)
                h = F.linear(h, w, lin.bias)
            else:
                h = mod(h)
        lin, w = next(it)
        return F.linear(h, w, lin.bias).squeeze(-1)

    def _scores_gpu(self, traj: torch.Tensor, with_input_grad_norm: bool):
        # two library calls: spectral normalisation of every layer, then the whole MLP
        layers = self._spectral_layers()
        weights = ops.spectral_normalize([m.weight_orig for m in layers], [m.weight_u for m in layers],
                                         [m.weight_v for m in layers], self.training, _EPS)
        slope, drop = self._activation()
        return ops.critic_scores(traj, list(weights), [m.bias for m in layers], slope, drop, self.training,
                                 with_input_grad_norm)

    def scores_and_input_grad_norm(self, traj: torch.Tensor):
        """``(score [B], ||d score_i / d traj_i||^2 [B])`` from one forward pass: the two quantities the
        critic loss needs of the real samples (hinge term and R1 penalty, ``models/hivt.py:244-252``)."""
        if traj.is_cuda:
            return self._scores_gpu(traj, True)
        r = traj.detach().clone().requires_grad_(True)
        score = self(r)
        (grad,) = torch.autograd.grad(outputs=score.sum(), inputs=r, create_graph=True, retain_graph=True,
                                      only_inputs=True)
        return score, grad.pow(2).reshape(grad.size(0), -1).sum(dim=1)


class _HorizonCritic(BaseTrajectoryCritic):
    _layers = 3

    def __init__(self, horizon: int, input_dim: int = 2, hidden_dim: int = 256) -> None:
        self.horizon = horizon
        super().__init__(input_dim=horizon * input_dim, hidden_dim=hidden_dim, num_layers=self._layers)

    def forward(self, traj: torch.Tensor) -> torch.Tensor:
        return super().forward(traj[:, :self.horizon])

    def scores_and_input_grad_norm(self, traj: torch.Tensor):
        # the score does not depend on steps beyond the horizon, so neither does the gradient norm
        return super().scores_and_input_grad_norm(traj[:, :self.horizon])


class ShortScaleCritic(_HorizonCritic):
    """First ``horizon`` steps, 2 hidden layers (``short_scale_critic.py:11-29``)."""
    _layers = 2

    def __init__(self, horizon: int = 10, input_dim: int = 2, hidden_dim: int = 128) -> None:
        super().__init__(horizon, input_dim, hidden_dim)


class MidScaleCritic(_HorizonCritic):
    """First ``horizon`` steps, 3 hidden layers (``mid_scale_critic.py:11-26``)."""
    _layers = 3

    def __init__(self, horizon: int = 20, input_dim: int = 2, hidden_dim: int = 256) -> None:
        super().__init__(horizon, input_dim, hidden_dim)


class LongScaleCritic(_HorizonCritic):
    """Full horizon, 4 hidden layers (``long_scale_critic.py:11-26``)."""
    _layers = 4

    def __init__(self, horizon: int, input_dim: int = 2, hidden_dim: int = 256) -> None:
        super().__init__(horizon, input_dim, hidden_dim)


def scores_multi(jobs):
    """``jobs``: ``(critic, trajectories [B, T, D], with_input_grad_norm)`` -- several critic evaluations on GPU
    tensors as ONE library call (``ops.critic_scores_multi``), with the spectral normalisation of all critics of a
    round in one launch.  A critic that appears k times goes through k power iterations in the order of its
    jobs, as k calls of its ``forward`` would.  Returns per job ``scores`` or ``(scores, grad_norm2)``."""
    rounds, seen = [], {}
    for idx, (critic, _, _) in enumerate(jobs):
        k = seen.get(id(critic), 0)
        seen[id(critic)] = k + 1
        while len(rounds) <= k:
            rounds.append([])
        rounds[k].append(idx)
    weights_of = {}
    same_members = len(rounds) > 1 and all([id(jobs[i][0]) for i in m] == [id(jobs[i][0]) for i in rounds[0]]
                                           for m in rounds[1:])
    layers0 = [m for idx in rounds[0] for m in jobs[idx][0]._spectral_layers()]
    if same_members and len(layers0) <= 16:
        # every critic evaluated len(rounds) times: all power iterations and normalisations as one node of the graph
        per_round = ops.spectral_normalize_rounds([m.weight_orig for m in layers0], [m.weight_u for m in layers0],
                                                  [m.weight_v for m in layers0], len(rounds),
                                                  jobs[rounds[0][0]][0].training, _EPS)
        for members, ws in zip(rounds, per_round):
            at = 0
            for idx in members:
                n = len(jobs[idx][0]._spectral_layers())
                weights_of[idx] = ws[at:at + n]
                at += n
        rounds = []
    for members in rounds:
        layers = [(idx, m) for idx in members for m in jobs[idx][0]._spectral_layers()]
        training = jobs[members[0]][0].training
        for start in range(0, len(layers), 16):
            part = layers[start:start + 16]
            ws = ops.spectral_normalize([m.weight_orig for _, m in part], [m.weight_u for _, m in part],
                                        [m.weight_v for _, m in part], training, _EPS)
            for (idx, _), w in zip(part, ws):
                weights_of.setdefault(idx, []).append(w)
    passes = []
    for idx, (critic, traj, with_r1) in enumerate(jobs):
        # the whole trajectory tensor goes down: a pass reads the leading columns its first layer is wide (the
        # critic's horizon), and passes over one tensor share its gradient
        passes.append((traj, weights_of[idx], [m.bias for m in critic._spectral_layers()], with_r1))
    first = jobs[0][0]
    slope, drop = first._activation()
    return ops.critic_scores_multi(passes, slope, drop, first.training)


# the critics' passes of a loss term as one library call (False: one call per critic and pass)
MULTI = os.environ.get("HIVT_CRITICS_MULTI", "1") != "0"


def multi_supported(critics, *tensors) -> bool:
    """Whether ``scores_multi`` applies: GPU tensors, plain trajectory critics that share the activation slope,
    the dropout rate and the training mode."""
    critics = list(critics)
    if not MULTI or not critics or not all(t.is_cuda for t in tensors):
        return False
    if not all(isinstance(c, BaseTrajectoryCritic) for c in critics):
        return False
    return len({(c._activation(), c.training) for c in critics}) == 1
