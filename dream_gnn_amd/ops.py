"""Torch-facing wrappers over the C ABI (``include/dgmi.h``).

PyTorch is plumbing here: it owns device memory and the stream; the arithmetic is
``libdgmi.so``.  Every op requires HIP ("cuda") tensors and raises otherwise.

Boundary being replaced (reference ``/root/reference/layers.py``):
  * ``graph.update_all(fn.copy_u('h','m'), fn.sum('m','h'))``  — layers.py:229-232
  * ``th.spmm(adj, support)``                                   — layers.py:312
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_L = _lib.lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "dream_gnn_amd ops run on the MI355X only: got a %s tensor. There is no CPU path "
                "(the CPU restatement under oracle/ is test infrastructure)." % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("tensors on different devices: %s vs %s" % (dev, t.device))
    return dev


def _check(t, dtype, name, ndim=None):
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must be %d-D, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def csr_from_coo(row: torch.Tensor, col: torch.Tensor, n_rows: int,
                 check_range: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Stable COO -> CSR on the device: ``(indptr[n_rows+1], indices[E], eid[E])``, all int32.

    ``eid`` is the stable permutation (``argsort(row, kind='stable')``); duplicates are kept.
    Replaces DGL's COO->CSR behind ``dgl.heterograph`` (data_loader.py:448, augmentation.py:65).
    ``check_range=True`` reads the kernel's error flag back (one host sync) and raises on an
    out-of-range row id.
    """
    dev = _require_device(row, col)
    _check(row, torch.int32, "row", 1)
    _check(col, torch.int32, "col", 1)
    if row.shape != col.shape:
        raise RuntimeError("row/col length mismatch")
    E = row.shape[0]
    with torch.cuda.device(dev):
        indptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        indices = torch.empty(E, dtype=torch.int32, device=dev)
        eid = torch.empty(E, dtype=torch.int32, device=dev)
        import ctypes
        need = ctypes.c_size_t(0)
        _lib.check(_L.dgmi_csr_from_coo_i32(_ptr(row), _ptr(col), E, n_rows, None, None, None, None,
                                            ctypes.byref(need), None), "dgmi_csr_from_coo_i32(size query)")
        ws = torch.empty(max(int(need.value), 256), dtype=torch.uint8, device=dev)
        have = ctypes.c_size_t(ws.numel())
        _lib.check(_L.dgmi_csr_from_coo_i32(_ptr(row), _ptr(col), E, n_rows, _ptr(indptr), _ptr(indices),
                                            _ptr(eid), _ptr(ws), ctypes.byref(have), _stream(dev)),
                   "dgmi_csr_from_coo_i32")
        if check_range:
            if int(ws[:4].view(torch.int32).item()) != 0:
                raise RuntimeError("csr_from_coo: a row id is outside [0, %d)" % n_rows)
    return indptr, indices, eid


def gather_f32(values: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    dev = _require_device(values, perm)
    _check(values, torch.float32, "values", 1)
    _check(perm, torch.int32, "perm", 1)
    out = torch.empty(perm.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_L.dgmi_gather_f32(_ptr(values), _ptr(perm), perm.shape[0], _ptr(out), _stream(dev)),
                   "dgmi_gather_f32")
    return out


def spmm_csr_raw(indptr, indices, vals, X, src_scale=None, dst_scale=None, out=None) -> torch.Tensor:
    """One ``dgmi_spmm_csr_f32`` call, no autograd.  X may be a row-strided 2-D view."""
    dev = _require_device(indptr, indices, vals, X, src_scale, dst_scale, out)
    _check(indptr, torch.int32, "indptr", 1)
    _check(indices, torch.int32, "indices", 1)
    if X.dtype != torch.float32 or X.dim() != 2:
        raise RuntimeError("X must be a 2-D float32 tensor, got %s %s" % (X.dtype, tuple(X.shape)))
    if X.stride(1) != 1 or (X.shape[0] > 1 and X.stride(0) < X.shape[1]):
        X = X.contiguous()
    n_src, F = X.shape
    ldx = X.stride(0) if n_src > 1 else max(F, 1)
    n_dst = indptr.shape[0] - 1
    if vals is not None:
        _check(vals, torch.float32, "vals", 1)
        if vals.shape[0] != indices.shape[0]:
            raise RuntimeError("vals/indices length mismatch")
    if src_scale is not None:
        src_scale = src_scale.reshape(-1)
        _check(src_scale, torch.float32, "src_scale", 1)
        if src_scale.shape[0] != n_src:
            raise RuntimeError("src_scale has %d entries, X has %d rows" % (src_scale.shape[0], n_src))
    if dst_scale is not None:
        dst_scale = dst_scale.reshape(-1)
        _check(dst_scale, torch.float32, "dst_scale", 1)
        if dst_scale.shape[0] != n_dst:
            raise RuntimeError("dst_scale has %d entries, graph has %d rows" % (dst_scale.shape[0], n_dst))
    if out is None:
        out = torch.empty((n_dst, F), dtype=torch.float32, device=dev)
    else:
        _check(out, torch.float32, "out", 2)
        if tuple(out.shape) != (n_dst, F):
            raise RuntimeError("out has shape %s, expected %s" % (tuple(out.shape), (n_dst, F)))
    with torch.cuda.device(dev):
        _lib.check(_L.dgmi_spmm_csr_f32(_ptr(indptr), _ptr(indices), _ptr(vals), _ptr(X), ldx, _ptr(src_scale),
                                        _ptr(dst_scale), _ptr(out), max(F, 1), n_dst, n_src, F, _stream(dev)),
                   "dgmi_spmm_csr_f32")
    return out


class CSRGraph:
    """A relation slice / sparse adjacency in the layout the kernels read.

    Holds the destination-major CSR (forward: ``Y = A X``) and, built lazily on first
    backward, the source-major CSR of the reversed edges (``dX = A^T dY``), both made by the
    device COO->CSR.  ``vals`` (optional) are per-edge values in the caller's COO order.
    Row = destination, col = source, as in ``th.spmm(adj, x)`` where ``adj[dst, src]``.
    """

    def __init__(self, dst: torch.Tensor, src: torch.Tensor, n_dst: int, n_src: int,
                 vals: Optional[torch.Tensor] = None, check_range: bool = True):
        _require_device(dst, src, vals)
        self.n_dst, self.n_src = int(n_dst), int(n_src)
        self._dst = dst.to(torch.int32).contiguous()
        self._src = src.to(torch.int32).contiguous()
        self._coo_vals = None if vals is None else vals.to(torch.float32).contiguous()
        if check_range and self._src.numel():
            lo, hi = int(self._src.min()), int(self._src.max())
            if lo < 0 or hi >= self.n_src:
                raise RuntimeError("source id out of range [0, %d): min %d max %d" % (self.n_src, lo, hi))
        self.indptr, self.indices, self.eid = csr_from_coo(self._dst, self._src, self.n_dst,
                                                           check_range=check_range)
        self.vals = None if self._coo_vals is None else gather_f32(self._coo_vals, self.eid)
        self._t = None

    @property
    def nnz(self) -> int:
        return int(self.indices.shape[0])

    @property
    def device(self):
        return self.indptr.device

    def transposed(self):
        """(indptr_t, indices_t, vals_t): CSR of the reversed edges, rows = source nodes."""
        if self._t is None:
            indptr_t, indices_t, eid_t = csr_from_coo(self._src, self._dst, self.n_src)
            vals_t = None if self._coo_vals is None else gather_f32(self._coo_vals, eid_t)
            self._t = (indptr_t, indices_t, vals_t)
        return self._t


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, g: CSRGraph, src_scale, dst_scale):
        ctx.g = g
        ctx.save_for_backward(src_scale, dst_scale)
        return spmm_csr_raw(g.indptr, g.indices, g.vals, X, src_scale, dst_scale)

    @staticmethod
    def backward(ctx, dY):
        src_scale, dst_scale = ctx.saved_tensors
        g = ctx.g
        dX = None
        if ctx.needs_input_grad[0]:
            indptr_t, indices_t, vals_t = g.transposed()
            # dX = diag(src_scale) A^T diag(dst_scale) dY : the same kernel on the reversed
            # edges with the two scales swapped.
            dX = spmm_csr_raw(indptr_t, indices_t, vals_t, dY.contiguous(), dst_scale, src_scale)
        return dX, None, None, None


def spmm_csr(g: CSRGraph, X: torch.Tensor, src_scale: Optional[torch.Tensor] = None,
             dst_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Y = diag(dst_scale) · A · diag(src_scale) · X`` with autograd w.r.t. ``X`` only.

    The reference needs no other gradient: adjacency values are constants (utils.py:24-27,
    augmentation.py:124) and ``ci``/``cj`` are non-learnable node data (data_loader.py:487-488).
    """
    if src_scale is not None and src_scale.requires_grad or dst_scale is not None and dst_scale.requires_grad:
        raise RuntimeError("spmm_csr: gradients w.r.t. the diagonal scales are not part of the path")
    return _SpMM.apply(X, g, src_scale, dst_scale)
