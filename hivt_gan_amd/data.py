"""PyG-free scene container, batching and a seeded nuScenes-shaped scene generator.

The reference stores one scene per ``TemporalData`` (``utils.py:8-62``, a
``torch_geometric.data.Data`` subclass) and batches scenes with
``Batch.from_data_list`` (``datasets/nuscenes_dataset.py:116-118``).  This module
provides the same attribute/item access and the same batching offsets without
torch_geometric:

* tensors are concatenated on dim 0, except ``edge_index`` and
  ``lane_actor_index`` which are concatenated on dim 1 (``utils.py:42-46``);
* ``edge_index`` is offset by the running node count, ``lane_actor_index`` by
  ``[[lanes so far], [nodes so far]]`` (``utils.py:48-62``);
* ``agent_index`` / ``av_index`` become global node ids (PyG offsets every key
  containing "index" by ``num_nodes``), 0-d ``theta`` becomes ``[B]``,
  ``origin[1,2]`` becomes ``[B,2]``, strings become lists.

The generator follows the tensor shapes, dtypes and edge orders produced by the
reference's offline preprocessor (``nuscenes_dataset.py:101-173,217-245,284-323``).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Tuple

import torch

_DIM1_KEYS = ("edge_index", "lane_actor_index")


class TemporalData:
    """Attribute- and item-addressable bag of tensors describing one scene or a batch."""

    def __init__(self, **fields) -> None:
        object.__setattr__(self, "_store", {})
        for k, v in fields.items():
            self._store[k] = v

    # -- mapping / attribute protocol (mirrors torch_geometric.data.Data) --------------
    def __getitem__(self, key: str):
        return self._store[key]

    def __setitem__(self, key: str, value) -> None:
        self._store[key] = value

    def __contains__(self, key: str) -> bool:
        return key in self._store

    def __getattr__(self, key: str):
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __setattr__(self, key: str, value) -> None:
        self._store[key] = value

    def keys(self) -> Iterable[str]:
        return self._store.keys()

    def get(self, key: str, default=None):
        return self._store.get(key, default)

    @property
    def num_graphs(self) -> int:
        return int(self._store.get("_num_graphs", 1))

    def to(self, device, non_blocking: bool = False) -> "TemporalData":
        out = TemporalData()
        for k, v in self._store.items():
            out._store[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def clone(self) -> "TemporalData":
        out = TemporalData()
        for k, v in self._store.items():
            out._store[k] = v.clone() if torch.is_tensor(v) else v
        return out

    def __repr__(self) -> str:
        parts = []
        for k, v in self._store.items():
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return "TemporalData(" + ", ".join(parts) + ")"


def _is_all_pairs(scene: TemporalData) -> bool:
    """Whether ``edge_index`` is every ordered pair (j, i), j != i, of the scene's agents in source-major
    order (``itertools.permutations``, nuscenes_dataset.py:139-141)."""
    n = int(scene.num_nodes)
    ei = scene["edge_index"]
    if ei.shape[1] != n * (n - 1):
        return False
    ar = torch.arange(n)
    src, dst = ar.repeat_interleave(n), ar.repeat(n)
    keep = src != dst
    return bool(torch.equal(ei[0], src[keep]) and torch.equal(ei[1], dst[keep]))


def collate(scenes: List[TemporalData]) -> TemporalData:
    """Batch scenes with the offsets of ``Batch.from_data_list`` for this data class."""
    if len(scenes) == 0:
        raise ValueError("cannot collate an empty list of scenes")
    out: Dict[str, object] = {}
    node_off, lane_off = 0, 0
    node_offs, lane_offs = [], []
    for s in scenes:
        node_offs.append(node_off)
        lane_offs.append(lane_off)
        node_off += int(s.num_nodes)
        lane_off += int(s.lane_vectors.shape[0]) if "lane_vectors" in s else 0
    keys = list(scenes[0].keys())
    for k in keys:
        vals = [s[k] for s in scenes]
        v0 = vals[0]
        if k == "num_nodes":
            out[k] = node_off
        elif k == "edge_index":
            out[k] = torch.cat([v + o for v, o in zip(vals, node_offs)], dim=1)
        elif k == "lane_actor_index":
            out[k] = torch.cat(
                [v + torch.tensor([[lo], [no]], dtype=v.dtype) for v, lo, no in zip(vals, lane_offs, node_offs)],
                dim=1)
        elif k in ("agent_index", "av_index"):
            out[k] = torch.tensor([int(v) + o for v, o in zip(vals, node_offs)], dtype=torch.long)
        elif torch.is_tensor(v0):
            out[k] = torch.stack(vals) if v0.dim() == 0 else torch.cat(vals, dim=0)
        elif isinstance(v0, (int, float)):
            out[k] = torch.tensor(vals)
        else:
            out[k] = list(vals)
    batch_vec = torch.cat([torch.full((int(s.num_nodes),), i, dtype=torch.long) for i, s in enumerate(scenes)])
    out["batch"] = batch_vec
    # per agent: first agent of its scene and one past its last (what the device-side edge builder walks),
    # and whether every scene's candidate list is all ordered pairs, as the reference's preprocessor writes
    # them (nuscenes_dataset.py:139-141) -- only then may the builder stand in for the list
    out["scene_range"] = torch.cat([torch.tensor([[o, o + int(s.num_nodes)]], dtype=torch.int32).expand(int(s.num_nodes), 2)
                                    for o, s in zip(node_offs, scenes)]).contiguous()
    out["all_pairs"] = all(_is_all_pairs(s) for s in scenes)
    out["_num_graphs"] = len(scenes)
    return TemporalData(**out)


# --------------------------------------------------------------------------------------
# Synthetic scenes
# --------------------------------------------------------------------------------------
def _rand_int(gen: torch.Generator, lo: int, hi: int) -> int:
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def make_scene(index: int,
               historical_steps: int = 20,
               future_steps: int = 30,
               local_radius: float = 50.0,
               num_agents: Optional[int] = None,
               num_lanes: Optional[int] = None,
               all_valid: bool = False,
               random_lane_attrs: bool = False,
               base_seed: int = 2022) -> TemporalData:
    """One synthetic scene; deterministic in ``index`` (generator seed ``base_seed + index``).

    Agents are constant-velocity tracks (speed U[0,15] m/s, dt 0.5 s, heading U[0,2pi)) starting in
    [-60,60]^2 m, each observed on one contiguous window of the 50 steps.  Lanes are ~1 m segments
    laid along 8-16 random polylines in the same box.  Every derived tensor (padding rules, BOS mask,
    rotate angles, displacement features, edge orders, lane-actor pre-filter) is computed by the rules
    of the reference preprocessor (``nuscenes_dataset.py:101-173,217-245,284-323``).
    """
    T, Hf = historical_steps, future_steps
    S = T + Hf
    gen = torch.Generator().manual_seed(base_seed + index)
    n = num_agents if num_agents is not None else _rand_int(gen, 12, 28)
    nl = num_lanes if num_lanes is not None else _rand_int(gen, 200, 400)

    start = (torch.rand(n, 2, generator=gen) * 2 - 1) * 60.0
    speed = torch.rand(n, generator=gen) * 15.0
    heading = torch.rand(n, generator=gen) * (2 * math.pi)
    vel = torch.stack([speed * torch.cos(heading), speed * torch.sin(heading)], dim=-1)
    steps = torch.arange(S, dtype=torch.float32).view(1, S, 1)
    world = start.unsqueeze(1) + vel.unsqueeze(1) * (0.5 * steps)  # [n, S, 2]

    padding_mask = torch.ones(n, S, dtype=torch.bool)
    if all_valid:
        padding_mask[:] = False
    else:
        long_track = torch.rand(n, generator=gen) < 0.7
        for i in range(n):
            if bool(long_track[i]):
                first = _rand_int(gen, 0, T - 2)
                last = _rand_int(gen, T - 1, S - 1)
            else:
                first = _rand_int(gen, 0, T - 1)  # must be seen at least once in the history
                last = _rand_int(gen, first, S - 1)
            padding_mask[i, first:last + 1] = False
    world = torch.where(padding_mask.unsqueeze(-1), torch.zeros_like(world), world)

    # scene frame: origin / heading of the actor with the longest history (nuscenes_dataset.py:103-122)
    hist_counts = (~padding_mask[:, :T]).sum(dim=1)
    av_index = int(torch.argmax(hist_counts))
    hv = (~padding_mask[av_index, :T]).nonzero().squeeze(-1)
    if hv.numel() >= 2:
        origin = world[av_index, hv[-1]].clone()
        hd = world[av_index, hv[-1]] - world[av_index, hv[-2]]
        theta = torch.atan2(hd[1], hd[0])
    else:
        origin = world[av_index, hv[-1]].clone() if hv.numel() else torch.zeros(2)
        theta = torch.tensor(0.0)
    rot = torch.tensor([[math.cos(float(theta)), -math.sin(float(theta))],
                        [math.sin(float(theta)), math.cos(float(theta))]], dtype=torch.float32)
    pos = torch.matmul(world - origin, rot)

    # no future for agents unseen at the last history step (nuscenes_dataset.py:125)
    padding_mask[padding_mask[:, T - 1], T:] = True

    # heading of each agent from its last two valid history points (nuscenes_dataset.py:217-226)
    rotate_angles = torch.zeros(n)
    for i in range(n):
        idx = (~padding_mask[i, :T]).nonzero().squeeze(-1)
        if idx.numel() >= 2:
            v = pos[i, idx[-1]] - pos[i, idx[-2]]
            rotate_angles[i] = torch.atan2(v[1], v[0])
        else:
            padding_mask[i, T:] = True

    bos_mask = torch.zeros(n, T, dtype=torch.bool)
    bos_mask[:, 0] = ~padding_mask[:, 0]
    bos_mask[:, 1:T] = padding_mask[:, :T - 1] & ~padding_mask[:, 1:T]

    # displacement features (nuscenes_dataset.py:228-245)
    feat = pos.clone()
    feat[:, T:] = torch.where((padding_mask[:, T - 1].unsqueeze(-1) | padding_mask[:, T:]).unsqueeze(-1),
                              torch.zeros(n, Hf, 2), feat[:, T:] - feat[:, T - 1].unsqueeze(-2))
    feat[:, 1:T] = torch.where((padding_mask[:, :T - 1] | padding_mask[:, 1:T]).unsqueeze(-1),
                               torch.zeros(n, T - 1, 2), feat[:, 1:T] - feat[:, :T - 1])
    feat[:, 0] = 0.0
    x, y = feat[:, :T].contiguous(), feat[:, T:].contiguous()

    # all ordered pairs, source-major (itertools.permutations order, nuscenes_dataset.py:139-141)
    ar = torch.arange(n)
    src = ar.repeat_interleave(n)
    dst = ar.repeat(n)
    off_diag = src != dst
    edge_index = torch.stack([src[off_diag], dst[off_diag]]).long().contiguous()

    # lanes: ~1 m segments along random polylines
    n_poly = _rand_int(gen, 8, 16)
    per = [nl // n_poly + (1 if j < nl % n_poly else 0) for j in range(n_poly)]
    lane_pos, lane_vec = [], []
    for cnt in per:
        if cnt == 0:
            continue
        p0 = (torch.rand(2, generator=gen) * 2 - 1) * 60.0
        h0 = float(torch.rand(1, generator=gen)) * 2 * math.pi
        curv = (float(torch.rand(1, generator=gen)) - 0.5) * 0.04
        ang = h0 + curv * torch.arange(cnt, dtype=torch.float32)
        seg = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1) * (0.9 + 0.2 * torch.rand(cnt, 1, generator=gen))
        pts = p0 + torch.cumsum(torch.cat([torch.zeros(1, 2), seg[:-1]]), dim=0)
        lane_pos.append(pts)
        lane_vec.append(seg)
    if lane_vec:
        lane_pos_t = torch.matmul(torch.cat(lane_pos) - origin, rot)
        lane_vectors = torch.matmul(torch.cat(lane_vec), rot)
    else:
        lane_pos_t = torch.zeros(0, 2)
        lane_vectors = torch.zeros(0, 2)
    L = lane_vectors.shape[0]
    if random_lane_attrs:
        is_inter = torch.randint(0, 2, (L,), generator=gen).to(torch.uint8)
        turn = torch.randint(0, 3, (L,), generator=gen).to(torch.uint8)
        control = torch.randint(0, 2, (L,), generator=gen).to(torch.uint8)
    else:  # the reference preprocessor writes zeros (nuscenes_dataset.py:284-286)
        is_inter = torch.zeros(L, dtype=torch.uint8)
        turn = torch.zeros(L, dtype=torch.uint8)
        control = torch.zeros(L, dtype=torch.uint8)

    # lane-major (lane, visible actor) pairs, pre-filtered by distance (nuscenes_dataset.py:304-322)
    visible = (~padding_mask[:, T - 1]).nonzero().squeeze(-1)
    if L > 0 and visible.numel() > 0:
        la_lane = torch.arange(L).repeat_interleave(visible.numel())
        la_actor = visible.repeat(L)
        node_pos = pos[visible, T - 1]
        la_vec = lane_pos_t.repeat_interleave(visible.numel(), 0) - node_pos.repeat(L, 1)
        keep = torch.norm(la_vec, dim=1) < local_radius
        lane_actor_index = torch.stack([la_lane[keep], la_actor[keep]]).long()
        lane_actor_vectors = la_vec[keep]
    else:
        lane_actor_index = torch.zeros(2, 0, dtype=torch.long)
        lane_actor_vectors = torch.zeros(0, 2)

    return TemporalData(
        x=x, positions=pos.clone(), edge_index=edge_index, y=y, num_nodes=n,
        padding_mask=padding_mask, bos_mask=bos_mask, rotate_angles=rotate_angles,
        lane_vectors=lane_vectors, is_intersections=is_inter, turn_directions=turn,
        traffic_controls=control, lane_actor_index=lane_actor_index,
        lane_actor_vectors=lane_actor_vectors, seq_id=f"synthetic-{base_seed + index}",
        av_index=av_index, agent_index=av_index, city="synthetic",
        origin=origin.unsqueeze(0), theta=theta)


def make_batch(num_scenes: int, first_index: int = 0, **scene_kwargs) -> TemporalData:
    """``num_scenes`` consecutive synthetic scenes collated into one batch (CPU tensors)."""
    return collate([make_scene(first_index + i, **scene_kwargs) for i in range(num_scenes)])


# --------------------------------------------------------------------------------------
# Fixed-shape batches
# --------------------------------------------------------------------------------------
class BatchSpec:
    """Tensor capacities of a fixed-shape batch.

    A training step whose tensor shapes never change needs no shape-dependent host work (allocation
    sizes, launch grids, read-backs of counts).  ``pad_batch`` brings every batch to these
    capacities; the padding is inert by construction:

    * padded agents are masked at every step (``padding_mask`` true), so the validity filter drops
      every edge that touches them and every loss weights them with zero;
    * padded agent-agent pairs are self loops of the last (always padded) agent;
    * padded lane-actor pairs carry a vector far outside any radius, so the radius filter drops them.

    ``edge_capacity`` bounds the three FILTERED edge lists (agent-agent over all history steps,
    lane-actor, last-step interaction); the kernels read the live counts from device memory.
    """

    def __init__(self, num_nodes: int, num_lanes: int, num_edges: int, num_lane_actor: int,
                 edge_capacity: Optional[tuple] = None) -> None:
        self.num_nodes = int(num_nodes)
        self.num_lanes = int(num_lanes)
        self.num_edges = int(num_edges)
        self.num_lane_actor = int(num_lane_actor)
        self.edge_capacity = tuple(int(c) for c in edge_capacity) if edge_capacity is not None else None

    @staticmethod
    def covering(batches: Iterable[TemporalData], slack: float = 1.05, multiple: int = 64) -> "BatchSpec":
        """Smallest spec (rounded up) that holds every batch of ``batches`` with ``slack`` head room."""
        def up(v: float) -> int:
            return int(-(-int(v * slack + 1) // multiple) * multiple)
        n = l = e = la = 0
        for b in batches:
            n = max(n, int(b.num_nodes))
            l = max(l, int(b["lane_vectors"].shape[0]))
            e = max(e, int(b["edge_index"].shape[1]))
            la = max(la, int(b["lane_actor_index"].shape[1]))
        return BatchSpec(up(n + 1), up(l), up(e), up(la))

    def key(self) -> tuple:
        """All capacities as one tuple (equal keys: the same padded shape)."""
        return (self.num_nodes, self.num_lanes, self.num_edges, self.num_lane_actor) + tuple(self.edge_capacity or ())

    @staticmethod
    def union(specs: Iterable["BatchSpec"]) -> "BatchSpec":
        """The smallest spec that holds whatever any of ``specs`` holds (capacity by capacity the largest)."""
        specs = list(specs)
        caps = [s.edge_capacity for s in specs if s.edge_capacity is not None]
        if caps and len(caps) != len(specs):
            raise ValueError("either every spec carries filtered-edge capacities or none does")
        return BatchSpec(max(s.num_nodes for s in specs), max(s.num_lanes for s in specs),
                         max(s.num_edges for s in specs), max(s.num_lane_actor for s in specs),
                         edge_capacity=tuple(max(c) for c in zip(*caps)) if caps else None)

    def padded_work(self) -> int:
        """What one step at this shape costs, up to a factor: the rows its kernels are launched over (padded rows
        are computed like real ones) -- filtered agent-agent and lane-actor edges, whose per-edge work dominates
        the step, plus the candidate lists the edge construction scans."""
        if self.edge_capacity is None:
            return self.num_nodes + self.num_edges + self.num_lane_actor
        return sum(self.edge_capacity[:3]) + self.num_nodes + (self.num_edges + self.num_lane_actor) // 8

    @staticmethod
    def shape_buckets(specs: List["BatchSpec"], max_buckets: int = 4, merge_equal: bool = True
                      ) -> Tuple[List["BatchSpec"], List[int]]:
        """At most ``max_buckets`` shapes for batches whose own smallest shapes are ``specs``: returns the bucket
        shapes and, for every entry of ``specs``, the index of its bucket (whose shape holds it).

        A step recorded at a fixed shape computes every padded row, so a batch should run at the smallest recorded
        shape that holds it; every recording costs memory and set-up time, so there are only a few.  The batches
        are ordered by ``padded_work`` and cut into consecutive runs; the cuts minimise the summed
        ``padded_work`` of the shapes the batches end up running at (dynamic programme over cut positions).
        ``merge_equal`` False: exactly ``min(max_buckets, len(specs))`` buckets even if some come out equal (ranks
        of a data-parallel job must all make the same number of recordings)."""
        n = len(specs)
        if n == 0:
            return [], []
        k = max(1, min(int(max_buckets), n))
        order = sorted(range(n), key=lambda i: (specs[i].padded_work(), specs[i].key()))
        cost = {}
        for a in range(n):  # cost[a, b]: batches order[a:b] in one bucket
            u = specs[order[a]]
            for b in range(a + 1, n + 1):
                u = BatchSpec.union([u, specs[order[b - 1]]])
                cost[a, b] = (b - a) * u.padded_work()
        inf = float("inf")
        best = [[inf] * (n + 1) for _ in range(k + 1)]  # best[j][b]: first b batches in exactly j buckets
        cut = [[0] * (n + 1) for _ in range(k + 1)]
        best[0][0] = 0
        for j in range(1, k + 1):
            for b in range(j, n + 1):
                for a in range(j - 1, b):
                    c = best[j - 1][a] + cost[a, b]
                    if c < best[j][b]:
                        best[j][b], cut[j][b] = c, a
        bounds, b = [], n
        for j in range(k, 0, -1):
            bounds.append((cut[j][b], b))
            b = cut[j][b]
        bounds.reverse()
        shapes, bucket_of, seen = [], [0] * n, {}
        for a, b in bounds:
            u = BatchSpec.union([specs[i] for i in order[a:b]])
            if merge_equal and u.key() in seen:
                idx = seen[u.key()]
            else:
                idx = len(shapes)
                shapes.append(u)
                seen.setdefault(u.key(), idx)
            for i in order[a:b]:
                bucket_of[i] = idx
        return shapes, bucket_of

    def __repr__(self) -> str:
        return (f"BatchSpec(num_nodes={self.num_nodes}, num_lanes={self.num_lanes}, num_edges={self.num_edges}, "
                f"num_lane_actor={self.num_lane_actor}, edge_capacity={self.edge_capacity})")


_FAR = 1.0e6  # metres; outside every radius


def _pad_rows(t: torch.Tensor, rows: int, value) -> torch.Tensor:
    out = torch.full((rows,) + tuple(t.shape[1:]), value, dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def pad_batch(batch: TemporalData, spec: BatchSpec, out: Optional[TemporalData] = None) -> TemporalData:
    """``batch`` brought to the capacities of ``spec`` (see ``BatchSpec`` for why the padding is inert).
    With ``out`` (a batch previously produced for the same spec) the tensors are written in place, so
    their addresses stay fixed."""
    n = int(batch.num_nodes)
    if n + 1 > spec.num_nodes:
        raise ValueError(f"batch has {n} agents; the spec needs at least one spare row beyond them "
                         f"(capacity {spec.num_nodes})")
    e, la, nl = batch["edge_index"].shape[1], batch["lane_actor_index"].shape[1], batch["lane_vectors"].shape[0]
    if e > spec.num_edges or la > spec.num_lane_actor or nl > spec.num_lanes:
        raise ValueError(f"batch exceeds {spec}: {e} pairs, {la} lane-actor pairs, {nl} lanes")
    last = spec.num_nodes - 1
    fields: Dict[str, object] = {}
    node_pad = {"x": 0.0, "positions": 0.0, "y": 0.0, "padding_mask": True, "bos_mask": False, "rotate_angles": 0.0}
    for k, v in node_pad.items():
        fields[k] = _pad_rows(batch[k], spec.num_nodes, v)
    for k in ("lane_vectors", "is_intersections", "turn_directions", "traffic_controls"):
        fields[k] = _pad_rows(batch[k], spec.num_lanes, 0)
    ei = torch.full((2, spec.num_edges), last, dtype=torch.long, device=batch["edge_index"].device)
    ei[:, :e] = batch["edge_index"]
    fields["edge_index"] = ei
    lai = torch.zeros((2, spec.num_lane_actor), dtype=torch.long, device=ei.device)
    lai[1] = last
    lai[:, :la] = batch["lane_actor_index"]
    fields["lane_actor_index"] = lai
    fields["lane_actor_vectors"] = _pad_rows(batch["lane_actor_vectors"], spec.num_lane_actor, _FAR)
    if "scene_range" in batch:  # padded agents belong to no scene: an empty range
        fields["scene_range"] = _pad_rows(batch["scene_range"], spec.num_nodes, 0)
    if out is None:
        out = TemporalData(**fields)
        out["num_nodes"] = spec.num_nodes
        out["_num_graphs"] = batch.num_graphs
        out["edge_capacity"] = spec.edge_capacity
        for k in ("agent_index", "av_index", "all_pairs"):
            if k in batch:
                out[k] = batch[k]
        return out
    for k, v in fields.items():
        out[k].copy_(v)
    for k in ("agent_index", "av_index", "all_pairs"):
        if k in batch:
            out[k] = batch[k]
    return out
