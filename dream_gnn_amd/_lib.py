"""ctypes binding of ``libdgmi.so`` (the C ABI declared in ``include/dgmi.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.  A missing
library is an error at import time: the product path has no fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must come first: libdgmi.so binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdgmi.so")
TORCH_LIB_PATH = os.path.join(_HERE, "libdgmi_torch.so")  # the dreamgnn_mi::* dispatcher ops over the C ABI

ABI_VERSION = 20

# name -> (restype, argtypes); mirrors include/dgmi.h one to one.
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_EPI = [ctypes.c_int32, ctypes.c_float, _vp, _i64, ctypes.c_float]  # act, act_slope, out_mask, ld_mask, out_mask_scale


SIGNATURES = {
    "dgmi_abi_version": (ctypes.c_int, []),
    "dgmi_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "dgmi_device_ok": (ctypes.c_int, []),
    "dgmi_csr_from_coo_i32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                             ctypes.POINTER(ctypes.c_size_t), _vp]),
    "dgmi_spmm_csr_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp, _vp, _vp, _i64, _i64,
                                         _i64, _i64] + _EPI + [_vp]),
    "dgmi_gather_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "dgmi_gather_concat_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "dgmi_csr_sliced_from_coo_i32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp, _vp, _vp,
                                                    ctypes.POINTER(ctypes.c_size_t), _vp]),
    "dgmi_csr_sliced_from_csr_i32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp, _vp, _vp,
                                                    ctypes.POINTER(ctypes.c_size_t), _vp]),
    "dgmi_spmm_sliced_planes_bytes": (ctypes.c_size_t, [_i64, ctypes.c_int32, _i64]),
    "dgmi_spmm_sliced_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp, _vp, _vp, _i64, _i64,
                                            _i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t] + _EPI
                             + [_vp]),
    "dgmi_knn_cosine_supported": (ctypes.c_int, [_i64, _i64, _i64]),
    "dgmi_knn_cosine_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, ctypes.c_int32]),
    "dgmi_knn_cosine_topk_f32": (ctypes.c_int, [_vp, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp, ctypes.c_size_t, _vp]),
    "dgmi_probe_row_gather_f32": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp]),
    "dgmi_epilogue_backward_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int32, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "dgmi_gather_add_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, ctypes.c_int32, _vp]),
    "dgmi_random_subset_workspace_bytes": (ctypes.c_size_t, []),
    "dgmi_random_subset_mask_f32": (ctypes.c_int, [_i64, _i64, ctypes.c_uint64, _vp, _vp, ctypes.c_size_t, _vp]),
    "dgmi_spmm_default_chunk": (ctypes.c_int32, [_i64, _i64]),
    "dgmi_spmm_plan_bytes": (ctypes.c_size_t, [_i64, _i64, ctypes.c_int32]),
    "dgmi_spmm_partials_bytes": (ctypes.c_size_t, [_i64, ctypes.c_int32, _i64]),
    "dgmi_spmm_plan_build": (ctypes.c_int, [_vp, _i64, _i64, ctypes.c_int32, _vp, ctypes.c_size_t, _vp,
                                            ctypes.POINTER(ctypes.c_size_t), _vp]),
    "dgmi_spmm_csr_planned_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp, _vp, _vp, _i64,
                                                 _i64, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp, ctypes.c_size_t] + _EPI
                                  + [_vp]),
    "dgmi_random_subset_select": (ctypes.c_int, [_i64, _i64, ctypes.c_uint64, ctypes.c_uint32, _vp, _vp, ctypes.c_size_t,
                                                 _vp]),
    "dgmi_random_subset_select_batch": (ctypes.c_int, [ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "dgmi_random_subset_select_batch_dseed": (ctypes.c_int, [ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "dgmi_keep_mask_f32": (ctypes.c_int, [_vp, ctypes.c_int32, _i64, _vp, _vp]),
    "dgmi_scale_rows_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "dgmi_weighted_colsum_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, ctypes.c_int32, _vp, _i64, _vp]),
    "dgmi_rank_add_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, ctypes.c_int32, _vp]),
    "dgmi_compact_layout_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "dgmi_compact_layout_i32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp,
                                               ctypes.c_size_t, _vp]),
    "dgmi_set_tuning": (ctypes.c_int, [ctypes.c_char_p, _i64]),
    "dgmi_row_multiplicity_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp, _vp, _vp]),
}


class DgmiError(RuntimeError):
    """A non-zero dgmi_status came back from the C ABI."""


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dream_gnn_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dream_gnn_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what dgmi.h declares
        fn.restype = res
        fn.argtypes = args
    got = lib.dgmi_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libdgmi.so ABI version %d != expected %d; rebuild" % (got, ABI_VERSION))
    return lib


lib = _load()


def _load_torch_ops():
    """Register the ``dreamgnn_mi`` operator library (csrc/dgmi_torch.cpp).  The product path calls the
    kernels through these ops; a missing library is an error, as for libdgmi.so."""
    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError(
            "dream_gnn_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C dream_gnn_amd/csrc`. There is no CPU fallback." % TORCH_LIB_PATH)
    torch.ops.load_library(TORCH_LIB_PATH)
    return torch.ops.dreamgnn_mi


torch_ops = _load_torch_ops()


def set_tuning(name: str, value: int) -> None:
    """Override one launch parameter of the library (``dgmi_set_tuning``; measurement tools and one test)."""
    check(lib.dgmi_set_tuning(name.encode(), int(value)), "dgmi_set_tuning(%s)" % name)


def check(status: int, what: str) -> None:
    if status != 0:
        raise DgmiError("%s failed: %s (status %d)" % (what, lib.dgmi_status_string(status).decode(), status))
