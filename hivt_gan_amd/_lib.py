"""ctypes binding of ``libhivt_mi355.so`` (the C ABI declared in ``include/hivt_mi355.h``).

The library is built in-tree by ``hivt_gan_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the shared library is missing the first call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhivt_mi355.so")

_P = c_void_p  # every device pointer crosses as void*

# name -> (restype, argtypes); order and meaning follow include/hivt_mi355.h
SIGNATURES = {
    "hivt_version": (c_int, []),
    "hivt_last_error": (c_char_p, []),
    "hivt_edges_workspace_bytes": (c_size_t, [c_int64]),
    "hivt_edges_temporal": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32,
                                    c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hivt_edges_radius": (c_int, [_P, c_int64, _P, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hivt_edges_all_pairs_workspace_bytes": (c_size_t, [c_int64]),
    "hivt_edges_all_pairs_csr": (c_int, [_P, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32, c_float,
                                         c_int64, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hivt_csr_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hivt_csr_build": (c_int, [_P, c_int64, _P, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "hivt_csr_build_static": (c_int, [_P, c_int64, _P, c_int32, _P, _P, _P]),
    "hivt_edge_gather_rotate_fwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int32, c_int64, _P, _P, _P, _P, _P]),
    "hivt_edge_gather_rotate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int64, _P, _P, _P, _P]),
    "hivt_edge_relpose_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "hivt_seg_attn_fwd": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int32, c_int64, c_int32,
                                  c_int32, c_float, c_uint64, _P, _P, _P, _P, c_int, _P]),
    "hivt_seg_attn_bwd": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, c_int32,
                                  c_int64, c_int32, c_int32, c_float, c_uint64, _P, _P, c_int64, _P, _P, c_int64,
                                  c_int32, c_int, _P]),
    "hivt_seg_attn_fwd_timed": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int32, c_int64,
                                        c_int32, c_int32, c_float, c_uint64, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "hivt_seg_attn_bwd_timed": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P,
                                        c_int32, c_int64, c_int32, c_int32, c_float, c_uint64, _P, _P, c_int64, _P,
                                        _P, c_int64, c_int32, c_int, _P, _P, _P]),
    "hivt_stack_pairs": (c_int, [_P, c_int32, _P]),
    "hivt_segment_sum_rows": (c_int, [_P, c_int64, c_int32, _P, _P, c_int32, _P, c_int64, _P]),
    "hivt_segment_sum_rows_half": (c_int, [_P, c_int64, c_int32, _P, _P, c_int32, _P, c_int64, _P]),
    "hivt_gated_update_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P]),
    "hivt_gated_update_bwd": (c_int, [_P, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, _P,
                                      _P, _P, c_int64, _P, c_int64, _P]),
    "hivt_linear_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int32, _P, _P, c_float, c_int, c_float,
                                c_float, c_uint64, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P,
                                c_int64, _P, _P]),
    "hivt_dropout_bwd": (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, c_uint64, _P, c_int64, c_int32, _P,
                                 c_int64, _P]),
    "hivt_dropout_bwd_half": (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, c_uint64, _P, c_int64, c_int32, _P,
                                      c_int64, _P]),
    "hivt_rowsum_by_code_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "hivt_rowsum_by_code": (c_int, [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "hivt_linear_bwd_data": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_float, _P,
                                     c_int64, _P, c_int64, _P]),
    "hivt_linear_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "hivt_linear_wgrad": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "hivt_linear_wgrad_half": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, c_int32,
                                       _P]),
    "hivt_linear_wgrad_group_workspace_bytes": (c_size_t, [_P, c_int32]),
    "hivt_linear_wgrad_group": (c_int, [_P, c_int32, _P, c_size_t, _P]),
    "hivt_ln_fwd": (c_int, [_P, c_int64, _P, _P, c_float, c_int64, c_int32, _P, c_int64, _P, _P]),
    "hivt_ln_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hivt_ln_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int, c_int64, c_int32, _P, c_int64, _P, c_int64,
                            _P, _P, _P, c_size_t, _P]),
    "hivt_ln_bwd_half": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int, c_int64, c_int32, _P, c_int64, _P, c_int64,
                                 _P, _P, _P, c_size_t, _P]),
    "hivt_linear_bwd_data_half": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_float, _P,
                                          c_int64, _P, c_int64, c_int32, _P]),
    "hivt_ln_bwd_partial_rows": (c_size_t, [c_int64]),
    "hivt_reduce_rows_group": (c_int, [_P, c_int32, _P]),
    "hivt_temporal_attn_fwd": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_uint64, _P, _P, _P, _P]),
    "hivt_temporal_attn_bwd": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_uint64, _P, _P,
                                       _P]),
    "hivt_temporal_attn_bwd_half": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_uint64, _P,
                                            _P, _P]),
    "hivt_laplace_workspace_bytes": (c_size_t, [c_int32]),
    "hivt_laplace_minmode_fwd": (c_int, [_P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P,
                                         _P, c_size_t, _P]),
    "hivt_laplace_minmode_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P, _P,
                                         _P, _P, _P]),
    "hivt_embed2_fwd": (c_int, [_P, c_int64, c_int32, _P, _P, _P, _P, c_float, c_int64, c_int32, _P, _P]),
    "hivt_embed2_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "hivt_embed2_bwd_partial_rows": (c_size_t, [c_int64, c_int32, c_int32]),
    "hivt_embed2_bwd": (c_int, [_P, _P, c_int64, c_int32, _P, _P, _P, _P, c_float, c_int64, c_int32, _P, _P, _P,
                                c_size_t, _P]),
    "hivt_critic_saved_floats": (c_size_t, [c_int64, c_int32, _P, c_int32]),
    "hivt_critic_workspace_bytes": (c_size_t, [c_int64, c_int32, _P]),
    "hivt_critic_fwd": (c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_float, c_float, c_uint64, _P, c_int32, _P,
                                _P, _P, _P]),
    "hivt_critic_bwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_int32, _P, _P, c_float, c_float, _P, _P, _P, _P, _P,
                                c_size_t, _P]),
    "hivt_critics_fwd": (c_int, [_P, c_int32, c_float, c_float, _P, _P]),
    "hivt_critics_bwd_workspace_bytes": (c_size_t, [_P, c_int32]),
    "hivt_critics_bwd": (c_int, [_P, c_int32, c_float, c_float, _P, c_size_t, _P]),
    "hivt_hinge_r1_fwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, _P, _P]),
    "hivt_hinge_r1_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, _P, _P, _P, _P]),
    "hivt_weighted_mean_fwd": (c_int, [_P, _P, _P, c_int64, c_float, _P, _P]),
    "hivt_weighted_mean_bwd": (c_int, [_P, _P, _P, c_int64, c_float, _P, _P]),
    "hivt_batch_prepare": (c_int, [_P, _P, _P, c_int64, c_int64, _P, c_int64, _P, _P, _P, c_int32, c_int32, c_int32,
                                   c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "hivt_row_subset_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "hivt_row_subset": (c_int, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64, _P, c_int64, _P, c_int64, _P,
                                _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hivt_center_rows": (c_int, [_P, _P, c_int64, _P, c_int32, c_int32, c_int32, _P, _P, _P]),
    "hivt_rows_put_tokens": (c_int, [_P, c_int64, _P, _P, c_int64, c_int32, c_int32, _P, _P]),
    "hivt_temporal_input_fwd": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P]),
    "hivt_gather_live_rows": (c_int, [_P, _P, _P, c_int32, c_int32, _P, _P]),
    "hivt_decoder_output_fwd": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_float, _P, _P]),
    "hivt_decoder_output_bwd": (c_int, [_P, _P, c_int64, c_int64, c_int32, c_int32, c_float, _P, _P, _P]),
    "hivt_category_table_fwd": (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "hivt_category_table_bwd": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P]),
    "hivt_select_mode_fwd": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "hivt_select_mode_bwd": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "hivt_sum_padded_rows": (c_int, [_P, _P, c_int32, c_int64, c_int32, _P, _P]),
    "hivt_hinge_r1_multi_fwd": (c_int, [_P, _P, _P, c_int32, _P, _P, c_int64, c_float, _P, _P]),
    "hivt_hinge_r1_multi_bwd": (c_int, [_P, _P, _P, c_int32, _P, _P, c_int64, c_float, _P, _P, _P, _P]),
    "hivt_generator_total_fwd": (c_int, [_P, _P, c_int32, _P, _P, c_int64, c_float, _P, _P]),
    "hivt_generator_total_bwd": (c_int, [_P, c_int32, _P, _P, c_int64, c_float, _P, _P, _P]),
    "hivt_set_matmul_bf16": (c_int, [c_int]),
    "hivt_spectral_norm_fwd": (c_int, [c_int32, _P, _P, _P, _P, _P, c_int32, c_float, _P, _P, _P, _P, _P]),
    "hivt_spectral_norm_bwd": (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hivt_step_guards": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P]),
    "hivt_adamw_step": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, _P, c_float, c_float, c_float, c_float, c_float, _P]),
    "hivt_adamw_update": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, _P, _P, c_float, c_float, c_float, c_float, c_float,
                                  _P]),
}

# entry points whose int return value is an answer rather than an error code
INT_VALUED = ("hivt_version", "hivt_set_matmul_bf16")

_lib = None


class HivtLibraryError(RuntimeError):
    """A C-ABI call returned a non-zero code; the message is ``hivt_last_error()``."""


def load() -> ctypes.CDLL:
    """Load the shared library once and declare every prototype."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HivtLibraryError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


_functions = {}
_fast = None  # the generated CPython binding of the same entry points, when it has been built


def _load_fast():
    """``_hivt_fast`` (built next to the library from ``csrc/gen_fastcall.py``): one METH_FASTCALL
    wrapper per entry point.  Same functions, same arguments; a call costs a fraction of a microsecond
    of host time instead of several with ctypes.  Absent module -> the ctypes binding serves."""
    global _fast
    if _fast is None:
        load()  # the shared library must be there either way: no fallback for the kernels
        try:
            from . import _hivt_fast
            _hivt_fast.set_error_class(HivtLibraryError)
            _fast = _hivt_fast
        except ImportError:
            _fast = False
    return _fast


def _function(name: str):
    fn = _functions.get(name)
    if fn is None:
        fast = _load_fast()
        if fast and hasattr(fast, name):
            fn = getattr(fast, name)
        else:
            cfn = getattr(load(), name)
            if cfn.restype is c_int and name not in INT_VALUED:
                def fn(*args, _cfn=cfn, _name=name):
                    rc = _cfn(*args)
                    if rc != 0:
                        msg = load().hivt_last_error().decode("utf-8", "replace")
                        raise HivtLibraryError(f"{_name} failed with code {rc}: {msg}")
            else:
                fn = cfn
        _functions[name] = fn
    return fn


def call(name: str, *args) -> None:
    """Invoke an ``int``-returning entry point; a non-zero return code raises ``HivtLibraryError``."""
    fn = _functions.get(name)
    if fn is None:
        fn = _function(name)
    fn(*args)


_sizes = {}


def query(name: str, *args) -> int:
    """Invoke a size query (``*_workspace_bytes``) or ``hivt_version``; answers are pure functions of the
    arguments and are remembered."""
    key = (name, args)
    size = _sizes.get(key)
    if size is None:
        if len(_sizes) > 4096:
            _sizes.clear()
        conv = [(ctypes.c_int32 * len(a))(*a) if isinstance(a, tuple) else a for a in args]  # tuples: int32 arrays
        size = _sizes[key] = int(_function(name)(*conv))
    return size


def value(name: str, *args) -> int:
    """Invoke an entry point that returns an integer answer (no caching, no error mapping)."""
    return int(_function(name)(*args))
