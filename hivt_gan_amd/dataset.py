"""Reader for preprocessed scenes on disk and a rank-sharded batch iterator.

The reference keeps one scene per ``<root>/<split>_processed/*.pt`` file, listed in sorted order and
optionally truncated to ``max_samples`` (``datasets/nuscenes_dataset.py:20-41``); every loaded scene
has its index / vector tensors brought to canonical shapes and malformed ones rejected with
``ValueError`` (``_sanitize``, ``nuscenes_dataset.py:55-103``); batches come from
``Batch.from_data_list`` (``:116-118``) through a loader with ``shuffle`` and a batch size
(``datamodules/nuscenes_datamodule.py:76-96``).

File format.  The reference pickles its ``torch_geometric`` data object whole, which can only be read
back by unpickling arbitrary classes with torch_geometric installed.  Files here hold the same fields
as a flat ``dict`` of tensors, numbers and strings, written by ``save_scene`` and read with
``torch.load(..., weights_only=True)``, so nothing in a file is executed.  A file in the reference's
pickled form is refused with an explanation instead of being unpickled.
"""
from __future__ import annotations

import os
from typing import Callable, Iterator, List, Optional

import torch

from .data import TemporalData, collate


def sanitize(data: TemporalData) -> TemporalData:
    """Canonical shapes of the index / vector fields (``nuscenes_dataset.py:55-103``):
    ``lane_actor_index`` and ``edge_index`` always ``[2, E]`` (an empty tensor becomes ``[2, 0]``, a lone
    edge stored as ``[2]`` becomes ``[2, 1]``), ``lane_actor_vectors`` ``[E, 2]`` and ``lane_vectors``
    ``[L, 2]`` (empty ones reshaped, non-tensors replaced by empty tensors); anything else raises
    ``ValueError``."""
    if "lane_actor_index" in data:
        lai = data["lane_actor_index"]
        if not torch.is_tensor(lai):
            data["lane_actor_index"] = torch.empty((2, 0), dtype=torch.long)
        elif lai.numel() == 0:
            data["lane_actor_index"] = lai.reshape(2, 0)
        elif lai.dim() == 1 and lai.size(0) == 2:
            data["lane_actor_index"] = lai.reshape(2, 1)
        elif lai.dim() != 2 or lai.size(0) != 2:
            raise ValueError(f"Bad lane_actor_index shape: {tuple(lai.shape)}")
    for key, what in (("lane_actor_vectors", "lane_actor_vectors"), ("lane_vectors", "lane_vectors")):
        if key in data:
            v = data[key]
            if not torch.is_tensor(v):
                data[key] = torch.empty((0, 2), dtype=torch.float)
            elif v.numel() == 0:
                data[key] = v.reshape(0, 2)
            elif v.dim() != 2 or v.size(-1) != 2:
                raise ValueError(f"Bad {what} shape: {tuple(v.shape)}")
    if "edge_index" in data:
        ei = data["edge_index"]
        if ei.numel() == 0:
            data["edge_index"] = ei.reshape(2, 0)
        elif ei.dim() == 1 and ei.size(0) == 2:
            data["edge_index"] = ei.reshape(2, 1)
        elif ei.dim() != 2 or ei.size(0) != 2:
            raise ValueError(f"Bad edge_index shape: {tuple(ei.shape)}")
    check_index_ranges(data)
    return data


def check_index_ranges(data: TemporalData) -> None:
    """Index fields must address existing agents / lanes: the device kernels index with them unchecked,
    where the reference's host-side indexing would raise.  Raises ``ValueError``."""
    n = int(data["num_nodes"]) if "num_nodes" in data else None
    if n is not None and "edge_index" in data and data["edge_index"].numel() > 0:
        ei = data["edge_index"]
        if int(ei.min()) < 0 or int(ei.max()) >= n:
            raise ValueError(f"edge_index entries must lie in [0, {n}); found [{int(ei.min())}, {int(ei.max())}]")
    if "lane_actor_index" in data and torch.is_tensor(data["lane_actor_index"]) and data["lane_actor_index"].numel() > 0:
        lai = data["lane_actor_index"]
        num_lanes = int(data["lane_vectors"].shape[0]) if "lane_vectors" in data else None
        if num_lanes is not None and (int(lai[0].min()) < 0 or int(lai[0].max()) >= num_lanes):
            raise ValueError(f"lane_actor_index lane ids must lie in [0, {num_lanes})")
        if n is not None and (int(lai[1].min()) < 0 or int(lai[1].max()) >= n):
            raise ValueError(f"lane_actor_index actor ids must lie in [0, {n})")


_PLAIN = (int, float, str, bool, type(None))


def save_scene(data: TemporalData, path: str) -> None:
    """One scene as a flat dict of tensors / numbers / strings (readable without executing anything)."""
    fields = {}
    for k in data.keys():
        v = data[k]
        if torch.is_tensor(v):
            fields[k] = v.detach().cpu()
        elif isinstance(v, _PLAIN):
            fields[k] = v
        else:
            raise TypeError(f"field {k!r} of type {type(v).__name__} cannot be stored")
    torch.save(fields, path)


def load_scene(path: str) -> TemporalData:
    """Reads a file written by ``save_scene`` and brings it to canonical shapes."""
    try:
        fields = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:  # a pickled object graph (the reference's own files) ends up here
        raise RuntimeError(
            f"{path}: not a flat dict of tensors.  Files pickled from a torch_geometric data object must be "
            f"converted once, where torch_geometric is installed, with "
            f"save_scene(TemporalData(**{{k: obj[k] for k in obj.keys}}), path)") from exc
    if not isinstance(fields, dict):
        raise RuntimeError(f"{path}: expected a dict of fields, found {type(fields).__name__}")
    return sanitize(TemporalData(**fields))


class SceneDataset:
    """``<root>/<split>_processed/*.pt`` in sorted order (``nuscenes_dataset.py:20-41,107-114``)."""

    def __init__(self, root: str, split: str = "train", transform: Optional[Callable] = None,
                 max_samples: Optional[int] = None) -> None:
        self.root = root
        self.split = split
        self.transform = transform
        self.processed_dir = os.path.join(root, f"{split}_processed")
        if not os.path.isdir(self.processed_dir):
            raise FileNotFoundError(f"Processed directory not found: {self.processed_dir}")
        names = sorted(f for f in os.listdir(self.processed_dir) if f.endswith(".pt"))
        self.processed_file_names: List[str] = names[:max_samples] if max_samples is not None else names

    def __len__(self) -> int:
        return len(self.processed_file_names)

    def __getitem__(self, idx: int) -> TemporalData:
        data = load_scene(os.path.join(self.processed_dir, self.processed_file_names[idx]))
        return self.transform(data) if self.transform is not None else data

    collate_fn = staticmethod(collate)


def batches(dataset, batch_size: int, shuffle: bool = False, seed: int = 0, epoch: int = 0, rank: int = 0,
            world_size: int = 1, drop_last: bool = False) -> Iterator[TemporalData]:
    """Collated batches of ``batch_size`` scenes for one epoch.  With ``world_size > 1`` every rank walks
    its own share of the (optionally shuffled, identically on all ranks) scene order; the order is padded
    by wrapping around so that all ranks see the same number of batches and enter the same collectives."""
    n = len(dataset)
    if n == 0:
        return
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    if world_size > 1:
        per_rank = -(-n // world_size)
        order = (order * (-(-per_rank * world_size // n)))[:per_rank * world_size]
        order = order[rank::world_size]
    for i in range(0, len(order), batch_size):
        chunk = order[i:i + batch_size]
        if drop_last and len(chunk) < batch_size:
            break
        yield collate([dataset[j] for j in chunk])
