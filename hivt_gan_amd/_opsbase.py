"""Helpers shared by the ``ops*`` modules: device addresses, stream handles, the row views the kernels take and the
error raised for shapes no kernel is built for."""
from __future__ import annotations

import contextlib
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _p(t: Optional[torch.Tensor], byte_offset: int = 0):
    """Device address as a plain integer (what a ``c_void_p`` parameter converts fastest from)."""
    if t is None:
        return None
    return t.data_ptr() + byte_offset


def _stream(t: torch.Tensor):
    """Raw handle of the calling thread's current stream on ``t``'s device."""
    return torch._C._cuda_getCurrentRawStream(t.device.index)


class UnsupportedShapeError(NotImplementedError):
    """A GPU tensor whose shape no kernel of the library is built for.  The library does not quietly hand
    such a tensor to the framework's operators: a model outside the built shapes (embed_dim above 128 for
    the LayerNorm kernels, a temporal sequence above 32 steps, ...) must fail visibly, not run a different
    code path."""


def _unsupported(what: str) -> "UnsupportedShapeError":
    return UnsupportedShapeError(f"hivt_gan_amd has no kernel for {what}; the library does not fall back to the "
                                 "framework's operators on GPU tensors")


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hivt_gan_amd kernels run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback in the product path")


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _ld(t: torch.Tensor) -> int:
    """Row stride of a 2-D tensor as a kernel wants it (a single row reports its width)."""
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A 2-D float32 view whose last dimension is contiguous (row stride arbitrary)."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"expected a 2-D tensor, got shape {tuple(t.shape)}")
    if (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) % 4 != 0) or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t


def _rows2(t: torch.Tensor) -> torch.Tensor:
    """2-D float32 view (leading dims flattened) whose last dimension has stride 1."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
