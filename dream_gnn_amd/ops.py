"""Graph-level layer over the ``dreamgnn_mi`` dispatcher ops (``csrc/dgmi_torch.cpp``), which in turn
are thin bindings of the C ABI (``include/dgmi.h``).

PyTorch is plumbing here: it owns device memory and the stream; the arithmetic is
``libdgmi.so``.  Every op requires HIP ("cuda") tensors and raises otherwise.  This module adds
what a functional op cannot hold: the layouts of one graph (CSR, transposed CSR, XCD-sliced, their
launch plans), built once and cached; the kernel choice per product; value and edge-dropout
views; autograd at the graph level.

Boundary being replaced (reference ``/root/reference/layers.py``):
  * ``graph.update_all(fn.copy_u('h','m'), fn.sum('m','h'))``  — layers.py:229-232
  * ``th.spmm(adj, support)``                                   — layers.py:312
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib

_L = _lib.lib          # ctypes view of the C ABI: host-only size / geometry queries
_T = _lib.torch_ops    # torch.ops.dreamgnn_mi: the dispatcher ops the kernels are launched through
_NULLCTX = contextlib.nullcontext()
#: output epilogue of a product: (act, slope, out_mask, mask_scale) — act 0 none / 1 leaky-relu(slope);
#: out_mask: 0/1 keep mask of the output (dropout), multiplied in with mask_scale = 1 / (1 - p)
_NO_EPI = (0, 0.0, None, 1.0)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device) -> int:
    """Raw hipStream_t of torch's current stream on ``device`` (kernels are enqueued there)."""
    return torch._C._cuda_getCurrentRawStream(device.index)


def _guard(device):
    """Make ``device`` current for the launch; free when it already is (the common case)."""
    if torch.cuda.current_device() == device.index:
        return _NULLCTX
    return torch.cuda.device(device)


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


def csr_from_coo(row: torch.Tensor, col: torch.Tensor, n_rows: int, n_cols: int = 0,
                 check_range: bool = False, return_flag: bool = False):
    """Stable COO -> CSR on the device: ``(indptr[n_rows+1], indices[E], eid[E])``, all int32.

    ``eid`` is the stable permutation (``argsort(row, kind='stable')``); duplicates are kept.
    Replaces DGL's COO->CSR behind ``dgl.heterograph`` (data_loader.py:448, augmentation.py:65).
    ``check_range=True`` reads the kernel's error flag back (one host sync) and raises if a row
    id is outside ``[0, n_rows)`` or, with ``n_cols > 0``, a column id outside ``[0, n_cols)``.
    ``return_flag=True`` appends that flag as a 1-element int32 device tensor instead (no sync), for
    callers that fold it into a readback of their own.  With the flag set the arrays are in bounds
    but meaningless: nothing may be derived from them.
    """
    _require_device(row, col)
    indptr, indices, eid, flag = _T.csr_from_coo(row, col, n_rows, n_cols)
    if check_range and int(flag.item()) != 0:
        raise RuntimeError("csr_from_coo: an id is outside [0, %d) x [0, %s)"
                           % (n_rows, n_cols if n_cols > 0 else "unchecked"))
    if return_flag:
        return indptr, indices, eid, flag
    return indptr, indices, eid


def gather_f32(values: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    _require_device(values, perm)
    return _T.gather_f32(values, perm)


class SpmmPlan:
    """nnz-balanced launch plan of one CSR (``dgmi_spmm_plan_build``): rows longer than
    ``chunk`` edges are cut into one-wave chunks.  Depends only on ``indptr``."""

    def __init__(self, indptr: torch.Tensor, nnz: int, chunk: Optional[int] = None):
        dev = _require_device(indptr)
        self.n_rows = int(indptr.shape[0] - 1)
        self.nnz = int(nnz)
        self.chunk = int(chunk) if chunk else int(_L.dgmi_spmm_default_chunk(self.n_rows, self.nnz))
        if int(_L.dgmi_spmm_plan_bytes(self.n_rows, self.nnz, self.chunk)) == 0:
            raise RuntimeError("invalid plan parameters (chunk=%d)" % self.chunk)
        self.buf = _T.plan_build(indptr, self.nnz, self.chunk)
        self._pbytes = {}

    def partials_bytes(self, F: int) -> int:
        b = self._pbytes.get(F)
        if b is None:
            b = self._pbytes[F] = int(_L.dgmi_spmm_partials_bytes(self.nnz, self.chunk, int(F)))
        return b

    def header(self):
        """(n_items, n_long_rows, n_slots, chunk, ...) — reads the device header back (tests)."""
        return tuple(int(v) for v in self.buf[:64].view(torch.int32).tolist())


def build_plan(indptr: torch.Tensor, nnz: int, chunk: Optional[int] = None) -> SpmmPlan:
    return SpmmPlan(indptr, nnz, chunk)


def _launch_spmm(dev, indptr, indices, vals, X, src_scale, dst_scale, out, plan, n_dst, n_src, F, ldx,
                 eid=None, keep=None, epi=None):
    """One ``dreamgnn_mi::spmm_csr_raw`` / ``spmm_csr_out`` dispatch (-> ``dgmi_spmm_csr_f32`` or
    ``dgmi_spmm_csr_planned_f32`` on torch's current stream; dtype / shape / device checks, output
    and scratch allocation happen in the op).  ``keep``: (n, 8) int32 subset descriptions
    (``random_subset_select``) applied through ``eid`` on the fly."""
    args = (indptr, indices, vals, eid if keep is not None else None, keep, X, src_scale, dst_scale,
            None if plan is None else plan.buf, 0 if plan is None else plan.chunk)
    epi = epi or _NO_EPI
    if out is None:
        return _T.spmm_csr_raw(*args, *epi)
    _T.spmm_csr_out(*args, out, *epi)
    return out


def _prep_scale(s, n, name):
    if s is None:
        return None
    if s.dim() != 1:
        s = s.reshape(-1)
    if s.dtype != torch.float32 or s.shape[0] != n or not s.is_contiguous():
        if s.dtype != torch.float32:
            raise RuntimeError("%s must be float32" % name)
        if s.shape[0] != n:
            raise RuntimeError("%s has %d entries, expected %d" % (name, s.shape[0], n))
        s = s.contiguous()
    return s


def spmm_csr_raw(indptr, indices, vals, X, src_scale=None, dst_scale=None, out=None,
                 plan: Optional[SpmmPlan] = None, eid=None, keep=None) -> torch.Tensor:
    """One SpMM launch, no autograd.  X may be a row-strided 2-D view.  ``plan=None``:
    ``dgmi_spmm_csr_f32`` (a wave per row); else ``dgmi_spmm_csr_planned_f32``.  ``keep`` (with
    ``eid``): subset descriptions for edge dropout on the fly."""
    dev = _require_device(indptr, indices, vals, X, src_scale, dst_scale, out, eid, keep)
    if keep is not None:
        keep = _prep_keep(keep)
    n_dst = indptr.shape[0] - 1
    if plan is not None and (plan.n_rows != n_dst or plan.nnz != indices.shape[0]):
        raise RuntimeError("plan was built for another CSR")
    src_scale = None if src_scale is None else src_scale.reshape(-1)
    dst_scale = None if dst_scale is None else dst_scale.reshape(-1)
    return _launch_spmm(dev, indptr, indices, vals, X, src_scale, dst_scale, out, plan, n_dst, X.shape[0],
                        X.shape[1] if X.dim() == 2 else 0, 0, eid, keep)


def _prep_keep(keep):
    """Subset descriptions as an (n, 8) contiguous int32 tensor."""
    if keep.dtype != torch.int32 or keep.dim() not in (1, 2) or keep.shape[-1] != 8:
        raise RuntimeError("keep must be int32 with 8 words per description, got %s %s" % (keep.dtype, tuple(keep.shape)))
    keep = keep.reshape(-1, 8)
    if keep.shape[0] > 8:
        raise RuntimeError("at most 8 subset descriptions per product")
    return keep if keep.is_contiguous() else keep.contiguous()


def _sliced_ok(X, out) -> bool:
    """16-B alignment requirements of ``dgmi_spmm_sliced_f32``."""
    if X.dtype != torch.float32 or X.stride(1) != 1 or X.stride(0) % 4 != 0 or X.data_ptr() % 16 != 0:
        return False
    return out is None or (out.is_contiguous() and out.data_ptr() % 16 == 0)


class SlicedCSR:
    """Source-sliced CSR for the XCD-local SpMM (``dgmi_csr_sliced_from_coo_i32``): edges sorted
    by (slice(src), row); ``segptr`` has ``n_slices * n_dst + 1`` entries."""

    N_SLICES = 8  # one slice of X per XCD

    def __init__(self, dst, src, n_dst, n_src, vals=None, n_slices: int = N_SLICES):
        _require_device(dst, src, vals)
        self.n_dst, self.n_src, self.n_slices = int(n_dst), int(n_src), int(n_slices)
        # range_flag: 1 if an id was out of range (device tensor; callers that build from unvalidated
        # edge lists read it — CSRGraph validates before it ever builds this layout)
        self.segptr, self.indices, self.eid, self.range_flag = _T.csr_sliced_from_coo(dst, src, self.n_dst, self.n_src,
                                                                                    self.n_slices)
        self.vals = None if vals is None else gather_f32(vals, self.eid)
        self.id_mult = False  # True: the id words carry integer multiplicities (bits 28..30 = m - 1), no value stream

    @classmethod
    def from_csr(cls, indptr, indices, eid, n_dst, n_src, n_slices: int = N_SLICES):
        """The same layout derived from the CSR of the same edge list (``dgmi_csr_sliced_from_csr_i32``): one
        stable partition pass by source slice instead of a full sort; bit-identical to the constructor."""
        self = cls.__new__(cls)
        self.n_dst, self.n_src, self.n_slices = int(n_dst), int(n_src), int(n_slices)
        self.segptr, self.indices, self.eid, self.range_flag = _T.csr_sliced_from_csr(indptr, indices, eid, self.n_src,
                                                                                    self.n_slices)
        self.vals = None
        self.id_mult = False
        return self

    def compacted(self, keep, vals=None, indices=None, id_mult=False) -> "SlicedCSR":
        """This layout with the edges dropped under the subset description(s) ``keep`` REMOVED
        (``dgmi_compact_layout_i32``: four streaming launches, no sort; survivors keep their order, so it is the layout
        a rebuild from the kept edge list would give — the reference's per-iteration construction, train.py:267 ->
        augmentation.py:48-65).  ``vals``: this layout's edge values (sliced order), compacted alongside; ``indices``
        (with ``id_mult``): id words to compact instead of the layout's own — the same ids carrying multiplicities.  The
        result runs the plain kernels: no ``eid`` stream, no hash per edge and pass, column passes as usual."""
        c = SlicedCSR.__new__(SlicedCSR)
        c.n_dst, c.n_src, c.n_slices = self.n_dst, self.n_src, self.n_slices
        c.segptr, c.indices, v = _T.compact_layout(self.segptr, self.indices if indices is None else indices, vals, self.eid,
                                                   _prep_keep(keep))
        c.vals = None if vals is None else v
        c.id_mult = bool(id_mult)
        c.eid, c.range_flag = None, self.range_flag  # positions no longer map to the caller's edge order
        return c

    _DEFAULT = object()

    def _prescale_pays(self, table_bytes: int, one_pass: bool) -> bool:
        """The pass moves 2 x table bytes (3.4e-13 s per byte measured), the in-kernel scale gather costs 2.7e-12 s
        per edge and column pass: pre-scale when the table is below ~8 bytes per edge-pass (config 4: 26-51 MB against
        10 M edges x 1-2 passes; a 400 MB table with 6 M edges keeps the in-kernel gather — the kernel-choice sweep)."""
        if table_bytes < PRESCALE_MIN_TABLE_BYTES:
            return False
        passes = 1 if one_pass or self.n_dst < 32768 or table_bytes // self.n_slices <= (4 << 20) else 2  # the launcher's column-pass rule
        return table_bytes <= 6 * int(self.indices.shape[0]) * passes

    def spmm(self, X, src_scale=None, dst_scale=None, out=None, vals=_DEFAULT, keep=None, epi=None, full_width=False,
             indices=None, id_mult=None):
        """``vals`` (in sliced order, see ``eid``) overrides the values given at construction; ``indices`` with
        ``id_mult=True``: id words carrying integer edge multiplicities (bits 28..30 = m - 1; then ``vals`` must be None);
        ``keep``: subset descriptions applied through ``eid`` (edge dropout on the fly); ``epi``: output
        epilogue (act, slope, out_mask, mask_scale) applied by the plane-reduce kernel; ``full_width``:
        never sweep the columns in two half-width passes (``column_passes = 1`` of the C ABI)."""
        vals = self.vals if vals is SlicedCSR._DEFAULT else vals
        if not X.is_cuda or X.device != self.segptr.device:
            _require_device(self.segptr, X)
        if X.dim() == 2 and X.shape[0] != self.n_src:
            raise RuntimeError("X has %d rows, the graph has %d source nodes" % (X.shape[0], self.n_src))
        if src_scale is not None and X.dim() == 2 and self._prescale_pays(X.shape[0] * X.shape[1] * 4, keep is not None or full_width):
            # diag(src_scale) X as ONE streaming pass, then the un-scaled gather: inside the kernel the scale is a
            # random 4-byte load per edge (and per column pass) — one more cache-line request beside the row's four —
            # which costs the config-4 products 8-18 % (tools/scale_cost_ab.py: 0.326 -> 0.285 ms, 0.298 -> 0.283 ms
            # including the pass)
            X = _T.scale_rows(X, src_scale.reshape(-1).contiguous())
            src_scale = None
        id_mult = self.id_mult if id_mult is None else bool(id_mult)
        args = (self.segptr, self.indices if indices is None else indices, vals, self.eid if keep is not None else None,
                None if keep is None else _prep_keep(keep), X, None if src_scale is None else src_scale.reshape(-1),
                None if dst_scale is None else dst_scale.reshape(-1), self.n_dst, self.n_slices)
        epi = epi or _NO_EPI
        passes = 1 if full_width else 0
        if out is None:
            return _T.spmm_sliced_raw(*args, *epi, passes, int(id_mult))
        _T.spmm_sliced_out(*args, out, *epi, passes, int(id_mult))
        return out


# XCD-local path is used when the gather table is a few L2s large (8 XCDs x 4 MiB): below, X is
# L2-hot anyway; far above, a 1/8 slice no longer fits an L2 and the plane traffic is pure cost.
# Measured on MI355X (tools/explore.py slicedcap, 10 M edges, F=128): sliced/planned time ratio
# 1.0 at a 6 MB table, 1.9-2.4x at 13-51 MB, 1.2x at 102 MB, 1.0 at 205 MB; 1.55x at average
# degree 100, 0.72x at 25 (a (row, slice) segment of 3 edges is all overhead).
FORCE_KERNEL = {"planned": "planned", "sliced": "sliced"}.get(os.environ.get("DGMI_FORCE_KERNEL", ""))
# Near-complete relation slices (f3, SURVEY §9-Q3: the reference's encoder graph holds EVERY train pair, both
# labels, data_loader.py:146-150,170 — lrssl shape: 464 897 of 519 603 cells) are handled one level up, as the
# COMPLEMENT form `column sum - complement cells` over the same CSR kernels (graph.py
# HeteroGraph.fused_relations_complement): 14 us against 40 us per relation-fused product at F = 344.  The round-2
# dense form (a materialised dense matrix through torch.mm) lost to the CSR kernel in context and was retired.
# Which products take the XCD-local form: fitted to a timing sweep of every form (round 3,
# tools/kernel_choice_sweep.py -> profiles/r03_kernel_choice_sweep.csv: F in {64, 128, 256, 344}, 20k .. 1.6M sources,
# average degree 16 .. 400, uniform and Zipf(1.1) degrees).  The sliced pair's advantage is set by the edges per
# (row, slice) segment — the average degree — and shrinks as the table outgrows what 8 L2s / the Infinity Cache hold:
#   degree >= 192: up to 1.5 GB tables     degree >= 96: up to 800 MB     degree >= 48: up to 420 MB
#   32 <= degree < 48: only where a slice is about one L2 (20 .. 80 MB tables: +7 .. 18 %)
# and tables below 6 MB are L2-hot for every kernel.  Long-row graphs (cut into virtual rows of SPLIT_ROW_EDGES) are
# judged by the degree of their virtual rows; their planned kernel is slower, so the split form pays from 4 MB tables
# and up to larger ones.  Round 2's narrower rule (10-160 MB at degree >= 64) picked a form > 10 % slower than the
# best on 63 of the sweep's 336 shapes; this one on 3 (all within 18 %, at the rule's edges).
SLICED_MIN_TABLE_BYTES = 6_000_000
SLICED_TIERS = ((48, 420_000_000), (96, 800_000_000), (192, 1_500_000_000))       # (average degree >=, table bytes <=)
SLICED_LOW_DEGREE = (32, 20_000_000, 80_000_000)                                    # degree >=, table bytes in [lo, hi]
SPLIT_MIN_TABLE_BYTES = 4_000_000
PRESCALE_MIN_TABLE_BYTES = 8_000_000  # XCD-local products on tables from this size scale X in a pass of its own
# Edge-dropped views of graphs that take the XCD-local form: the layout is COMPACTED once per view (a training step
# makes one view per edge list and runs each layout 3 x forward and 3 x backward) instead of evaluating keep(eid[p])
# per edge, product and column pass.  Config 4, ms per product un-dropped / dropped on the fly / dropped after
# compaction: see bench.py `variants`.  DGMI_COMPACT_DROPPED=0 keeps the on-the-fly kernels (A/B, tests).
COMPACT_DROPPED = os.environ.get("DGMI_COMPACT_DROPPED", "1") != "0"
# Weighted graphs in the XCD-local form whose values are `row scale x small integer` — every adjacency the reference
# builds: D^-1 (A + A^T + I), data_loader.py:297-308 + utils.py:11-17 — run without a value stream: the multiplicity rides
# in the id word's spare bits, the row scale goes where dst_scale (forward) / src_scale (transpose) go.  Found once per
# weighted graph by `dgmi_row_multiplicity_f32` (one host readback, never inside a stream capture); values that have no
# such form to within 2 ulp keep the value stream.  DGMI_MULT_FORM=0 switches it off (A/B, tests).
MULT_FORM = os.environ.get("DGMI_MULT_FORM", "1") != "0"
MULT_REL_TOL = 2.4e-7
MULT_SHIFT, MULT_MAX_IDS = 28, 1 << 28
COLUMN_PASS_MIN_ROWS, LONG_ROW_MIN_DEGREE = 32768, 384                # the launcher's column-pass rule (dgmi_sliced.hip); see _few_long_rows
SPLIT_MIN_DEGREE, SPLIT_DENSE_DEGREE = 8, 300                       # average degree (edges / rows) of the whole graph
SPLIT_MAX_TABLE_BYTES, SPLIT_MAX_TABLE_BYTES_DENSE = 350_000_000, 900_000_000


def _table_cap(tiers, degree: float) -> int:
    cap = 0
    for d, c in tiers:
        if degree >= d:
            cap = c
    return cap


class _Structure:
    """Everything about a CSRGraph that depends on the edge list only (shared by value views)."""

    __slots__ = ("n_dst", "n_src", "dst", "src", "indptr", "indices", "eid", "plan", "planned", "t",
                 "sliced", "sliced_t", "regular", "regular_t", "validated", "split", "split_t", "max_deg", "max_deg_t")


# Rows longer than this are cut into virtual rows before the XCD-local kernel sees them (power-law
# graphs): a (virtual row, slice) segment is then at most SPLIT_ROW_EDGES / 8 edges long.  A lane group
# streams 8 consecutive virtual rows of a slice as one dependent chain, so this length sets the critical
# path of the launch: Zipf(1.2), 10 M edges (tools/zipf_split_probe.py) 0.477 ms at 2048, 0.443 at 1024,
# 0.418 at 512 (with the launcher's column passes), 0.439 at 256 — in round 2.  Round 4 (tools/zipf_sweep.py, after the
# touch-ahead and with the light rows out of the sliced pass): 0.374 ms at 512 (23 632 virtual rows: below the 32 768 the
# launcher wants for its column passes), 0.329 at 256, 0.336 at 192, 0.363 at 128.
SPLIT_ROW_EDGES = 256


# Rows below this many edges do not go through the XCD-local kernel of a split graph at all: a (row, slice) segment of
# 0-5 edges is all bookkeeping.  A power-law graph is mostly such rows (Zipf(1.2), 10 M edges over 50 000 rows: 42 800 rows
# hold 5 % of the edges): they are gathered by the second-stage product instead, straight from the source table.
# tools/zipf_sweep.py, virtual rows of 256 edges: 0.348 ms with every row in the sliced pass, 0.329 / 0.334 / 0.336 with the
# rows below 24 / 48 / 96 edges out of it.
SPLIT_LIGHT_ROW_EDGES = 24


class _SplitSliced:
    """XCD-local product for a graph with extremely long rows.  Every HEAVY row is cut into virtual rows of at most
    ``SPLIT_ROW_EDGES`` edges (in CSR order), the sliced kernel runs on the virtual-row graph (regular by
    construction) and writes ``Yv``; a second, small product ``Y = C · [Yv ; Xs]`` then gives every row its sum:
    ``C[row, v] = 1`` adds a heavy row's virtual rows back together in order, and — **(r4)** when at least a quarter of
    the rows are LIGHT (< ``SPLIT_LIGHT_ROW_EDGES`` edges) — the light rows' own edges point into the second half of the
    same table, the (pre-scaled) source rows ``Xs``, so that they never become (row, slice) segments of one or two edges.
    ``dst_scale`` and the epilogue are applied by that second product."""

    def __init__(self, indptr, eid, cols_coo, n_rows, n_cols):
        dev = indptr.device
        nnz = int(eid.shape[0])
        deg = (indptr[1:] - indptr[:-1]).long()
        light = deg < SPLIT_LIGHT_ROW_EDGES
        n_light = int(light.sum())  # host syncs here and below: at construction only
        self.has_light = 4 * n_light >= n_rows and n_light < n_rows
        if not self.has_light:
            light = torch.zeros_like(light)
        nv = torch.where(light, torch.zeros_like(deg), torch.clamp((deg + SPLIT_ROW_EDGES - 1) // SPLIT_ROW_EDGES, min=1))
        vptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
        vptr[1:] = torch.cumsum(nv, 0)
        self.n_virtual = int(vptr[-1])
        rows_p = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg, output_size=nnz)
        rank = torch.arange(nnz, device=dev) - indptr.long()[rows_p]
        vrow_p = (vptr[rows_p] + rank // SPLIT_ROW_EDGES).to(torch.int32)
        cols_p = cols_coo[eid.long()]  # CSR order
        self.n_rows, self.n_cols = n_rows, n_cols
        if not self.has_light:
            vrow_coo = torch.empty(nnz, dtype=torch.int32, device=dev)
            vrow_coo[eid.long()] = vrow_p  # CSR position -> the caller's edge order
            self.sliced = SlicedCSR(vrow_coo, cols_coo, self.n_virtual, n_cols)
            self.c_indptr = vptr.to(torch.int32)
            self.c_indices = torch.arange(self.n_virtual, dtype=torch.int32, device=dev)
            self.c_eid = self.c_light_eid = self.c_order = None
        else:
            heavy_p = ~light[rows_p]
            # the heavy rows' edges, in CSR order (a stable sort by (slice, virtual row) keeps it inside a segment); the
            # layout's eid is composed back to the caller's edge order: values and dropout descriptions index that
            self.sliced = SlicedCSR(vrow_p[heavy_p].contiguous(), cols_p[heavy_p].contiguous(), self.n_virtual, n_cols)
            self.sliced.eid = eid[heavy_p][self.sliced.eid.long()].contiguous()
            # second stage: row r <- its virtual rows (ids < n_virtual) or, for a light row, its own edges (n_virtual + source)
            light_p = ~heavy_p
            ent_row = torch.cat([torch.repeat_interleave(torch.arange(n_rows, device=dev), nv, output_size=self.n_virtual), rows_p[light_p]])
            ent_idx = torch.cat([torch.arange(self.n_virtual, dtype=torch.int32, device=dev), cols_p[light_p] + self.n_virtual])
            order = torch.sort(ent_row, stable=True).indices
            self.c_order = order
            self.c_indices = ent_idx[order].contiguous()
            cptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
            cptr[1:] = torch.cumsum(torch.where(light, deg, nv), 0)
            self.c_indptr = cptr.to(torch.int32)
            self.c_light_eid = eid[light_p]  # the light edges' positions in the caller's order (values, dropout)
            # edge ids of the second stage's entries for dropout on the fly: virtual-row entries lie outside every description
            never = torch.full((self.n_virtual,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
            self.c_eid = torch.cat([never, self.c_light_eid])[order].contiguous()
        self.c_plan = build_plan(self.c_indptr, int(self.c_indices.shape[0]))

    def combine_values(self, coo_vals):
        """Values of the second stage's entries for a view with edge values ``coo_vals`` (the caller's order): 1 for a
        virtual row, the edge's own value for a light row's edge.  None when there is nothing to weight."""
        if coo_vals is None or not self.has_light:
            return None
        ones = torch.ones(self.n_virtual, dtype=torch.float32, device=coo_vals.device)
        return torch.cat([ones, coo_vals[self.c_light_eid.long()]])[self.c_order].contiguous()

    def spmm(self, X, src_scale, dst_scale, out, vals, keep=None, epi=None, compacted=None, c_vals=None):
        """``compacted``: the virtual-row layout with a view's dropped edges already removed (then ``vals`` is not
        consulted: it carries its own values); ``c_vals``: :meth:`combine_values` of the view."""
        F = X.shape[1]
        if self.has_light:
            # [Yv ; Xs] in ONE table: the sliced kernel gathers from its second half and writes its first half
            buf = torch.empty((self.n_virtual + self.n_cols, F), dtype=torch.float32, device=X.device)
            xs, yv = buf[self.n_virtual:], buf[:self.n_virtual]
            if src_scale is not None:
                torch.mul(X, src_scale.reshape(-1, 1), out=xs)
            else:
                xs.copy_(X)
            if compacted is not None:
                compacted.spmm(xs, None, None, yv)
            else:
                self.sliced.spmm(xs, None, None, yv, vals=vals, keep=keep)
            table, n_table, c_keep = buf, self.n_virtual + self.n_cols, keep
        else:
            if compacted is not None:
                yv = compacted.spmm(X, src_scale, None, None)
            else:
                yv = self.sliced.spmm(X, src_scale, None, None, vals=vals, keep=keep)
            table, n_table, c_keep = yv, self.n_virtual, None
        if out is None:
            out = torch.empty((self.n_rows, F), dtype=torch.float32, device=X.device)
        return _launch_spmm(X.device, self.c_indptr, self.c_indices, c_vals, table, None, dst_scale, out, self.c_plan,
                            self.n_rows, n_table, F, F, eid=self.c_eid, keep=c_keep, epi=epi)


class CSRGraph:
    """A relation slice / sparse adjacency in the layout the kernels read.

    Structure (shared, built once): the destination-major CSR (forward ``Y = A X``), lazily the
    source-major CSR of the reversed edges (``dX = A^T dY``) and the XCD-sliced layouts, each
    with its launch plan, all made by the device COO->CSR.  Values (per view): optional per-edge
    values in the caller's COO order, permuted on demand into each layout.  Row = destination,
    col = source, as in ``th.spmm(adj, x)`` where ``adj[dst, src]``.

    ``check_range=True`` costs one host sync (id range + maximum degree readback); pass ``False``
    for edge lists derived from an already validated graph.  :meth:`with_values` gives a view of
    the same structure with other edge values — how edge dropout is applied without re-sorting
    (a 0/1 keep mask as values).
    """

    def __init__(self, dst: torch.Tensor, src: torch.Tensor, n_dst: int, n_src: int,
                 vals: Optional[torch.Tensor] = None, check_range: bool = True, planned: bool = True,
                 regular: Optional[bool] = None, regular_t: Optional[bool] = None):
        _require_device(dst, src, vals)
        S = self._S = _Structure()
        S.n_dst, S.n_src = int(n_dst), int(n_src)
        S.dst = dst.to(torch.int32).contiguous()
        S.src = src.to(torch.int32).contiguous()
        if vals is not None and vals.shape[0] != S.dst.shape[0]:
            raise RuntimeError("vals/edge-list length mismatch")
        S.indptr, S.indices, S.eid, flag = csr_from_coo(S.dst, S.src, S.n_dst, S.n_src, return_flag=True)
        S.planned = planned
        S.t = S.sliced = S.sliced_t = S.split = S.split_t = None
        S.max_deg = S.max_deg_t = None  # longest row of the CSR / of the transposed CSR, once a readback has told
        S.validated = bool(check_range)
        # `regular`: no destination row is extremely long, so the XCD-local kernel (which walks a
        # (row, slice) segment sequentially) is safe to use.  Known after the one readback of a validated build
        # (below); for an unchecked build None = not known yet: decided — one readback of the maximum degree — at the
        # first product LARGE enough for the XCD-local form to be in question (a 4 MB table), never for the small
        # per-step graphs of the real datasets and never inside a stream capture.  (Until round 4 unchecked builds were
        # simply "not regular": every adjacency built by graph.similarity_graph / feature_similarity_graph — trusted by
        # construction — ran the planned kernel at scale, 0.84 ms against 0.41 ms per kNN-64 product.)
        S.regular = bool(regular) if regular is not None else None
        S.regular_t = regular_t
        self._set_values(None if vals is None else vals.to(torch.float32).contiguous())
        if check_range:
            # ids are checked BEFORE anything is derived from the CSR: the sort of an edge list with
            # ids outside [0, n) leaves indptr in bounds but meaningless, and a launch plan built
            # from it would be garbage (the plan kernels clamp, the product would still be wrong)
            self._validate(flag)
        S.plan = build_plan(S.indptr, int(S.indices.shape[0])) if planned else None

    # -- values -----------------------------------------------------------------------------------
    def _set_values(self, coo_vals, keep=None):
        self._coo_vals = coo_vals
        self._v = {}  # layout name -> values permuted into that layout
        self._c = {}  # layout name -> that layout with this view's dropped edges removed (SlicedCSR.compacted)
        self._keep = keep  # (n, 8) int32 subset descriptions applied on the fly, or None
        self._mask = None  # 0/1 keep mask already multiplied into the values by masked(), COO order
        self._vals_before_mask = None  # the values masked() multiplied its mask into (undropped() restores them)

    def with_values(self, coo_vals: Optional[torch.Tensor]) -> "CSRGraph":
        """A view sharing this graph's structure (and whatever it builds later) with other per-edge
        values, given in the original COO edge order."""
        if coo_vals is not None and coo_vals.shape[0] != self.nnz:
            raise RuntimeError("expected %d edge values, got %d" % (self.nnz, coo_vals.shape[0]))
        view = object.__new__(CSRGraph)
        view._S = self._S
        view._set_values(None if coo_vals is None else coo_vals.to(torch.float32).contiguous(), self._keep)
        return view

    def masked(self, keep: torch.Tensor) -> "CSRGraph":
        """View with edge e weighted by ``keep[e]`` (0/1), times the existing values if any."""
        keep = keep.to(torch.float32)
        view = self.with_values(keep if self._coo_vals is None else self._coo_vals * keep)
        view._keep = self._keep
        view._mask = keep if self._mask is None else self._mask * keep
        view._vals_before_mask = self._vals_before_mask if self._mask is not None else self._coo_vals
        view._v["mult"] = False  # zeros among the values: no scale x multiplicity form — and no per-step detection (a readback)
        return view

    def dropped(self, desc: torch.Tensor) -> "CSRGraph":
        """View of the same structure and values with edge dropout applied ON THE FLY: ``desc`` holds
        the 8-word description(s) of the surviving subset(s) (``random_subset_select``), evaluated
        per edge inside the kernels through each layout's ``eid`` — no mask is carried into the
        layouts, nothing is re-sorted, dropped edges are skipped (not multiplied by zero).

        On a view that already carries descriptions the new ones are ANDed in (an edge covered by
        several descriptions survives only if each keeps it, ``csrc/dgmi_keep.h``): the intersection
        of independently drawn subsets.  A chained dropout that must keep an EXACT count of the
        survivors (the reference's composition, augmentation.py:113-124) draws its subset among
        :meth:`survivors` instead — ``graph.random_edge_dropout_sparse`` does."""
        desc = _prep_keep(desc)
        view = object.__new__(CSRGraph)
        view._S = self._S
        view._coo_vals, view._v = self._coo_vals, self._v  # values (and their per-layout copies) are shared
        view._c = {}  # another set of survivors: its own compacted layouts
        view._mask, view._vals_before_mask = self._mask, self._vals_before_mask
        view._keep = desc if self._keep is None else torch.cat([self._keep, desc])
        if view._keep.shape[0] > 8:
            raise RuntimeError("at most 8 subset descriptions per graph view")
        return view

    def keep_mask(self) -> Optional[torch.Tensor]:
        """float 0/1 mask over the COO edge order of the on-the-fly dropout (None if there is none)."""
        return None if self._keep is None else keep_mask(self._keep, self.nnz)

    def survivors(self) -> Optional[torch.Tensor]:
        """float 0/1 mask over the COO edge order of every dropout this view carries — the on-the-fly
        descriptions and the masks folded into the values by :meth:`masked` — or None for an
        un-dropped graph."""
        m = self.keep_mask()
        if self._mask is not None:
            m = self._mask if m is None else m * self._mask
        return m

    def undropped(self) -> "CSRGraph":
        """The view of the same structure with every dropout removed (original values)."""
        view = object.__new__(CSRGraph)
        view._S = self._S
        vals = self._coo_vals
        if self._mask is not None:
            vals = self._vals_before_mask
        view._set_values(vals)
        return view

    def _compacted(self, name: str, sliced: "SlicedCSR", mult=None) -> Optional["SlicedCSR"]:
        """The XCD-sliced layout ``name`` with this view's dropped edges removed, made on first use and kept with the
        view (None when the view drops nothing or compaction is switched off).  ``mult``: the (row scale, id words)
        of :meth:`_mult_ids` — then the id words with their multiplicities are what is compacted, and no values."""
        if self._keep is None or not COMPACT_DROPPED:
            return None
        c = self._c.get(name)
        if c is None:
            if mult is not None:
                c = sliced.compacted(self._keep, None, indices=mult[2], id_mult=True)
            else:
                c = sliced.compacted(self._keep, self._vals_for(name, sliced.eid))
            self._c[name] = c
        return c

    def _mult_form(self):
        """``(kind, scale, code_coo[nnz])`` when this view's edge values are ``scale[node] * m`` with ``m`` an integer in
        1..8 (``code = m - 1``, COO order), else None.  ``kind == "row"``: ``scale`` is indexed by the destination (row)
        — the reference's ``normalize(adj + adj.T + I)`` adjacencies, ``D^-1 M``; ``kind == "col"``: by the source
        (column) — the TRANSPOSE of such an adjacency handed in as a graph of its own (``M D^-1``).  Decided once per
        weighted graph (shared by its dropped views): one kernel + one host readback per form tried."""
        if self._coo_vals is None or not MULT_FORM or self._S.n_src > MULT_MAX_IDS or self._S.n_dst > MULT_MAX_IDS:
            return None
        m = self._v.get("mult")
        if m is None:
            if self._capturing():
                return None  # the decision needs a readback: not inside a capture (the value stream is always right)
            S = self._S
            m = False
            scale, code, fail = _T.row_multiplicity(S.indptr, self.vals, MULT_REL_TOL)
            if int(fail.item()) == 0:
                code_coo = torch.empty_like(code)
                code_coo[S.eid.long()] = code
                m = ("row", scale, code_coo)
            else:
                indptr_t, _, vals_t, _ = self.transposed()
                scale, code, fail = _T.row_multiplicity(indptr_t, vals_t, MULT_REL_TOL)
                if int(fail.item()) == 0:
                    code_coo = torch.empty_like(code)
                    code_coo[S.t[2].long()] = code
                    m = ("col", scale, code_coo)
            self._v["mult"] = m
        return m or None

    def _mult_ids(self, name: str, sliced: "SlicedCSR"):
        """``(kind, scale, id words of layout `name` with the multiplicities in bits 28..30)`` or None."""
        mf = self._mult_form()
        if mf is None:
            return None
        ids = self._v.get(name + "/ids")
        if ids is None:
            ids = self._v[name + "/ids"] = sliced.indices | (mf[2][sliced.eid.long()] << MULT_SHIFT)
        return mf[0], mf[1], ids

    @staticmethod
    def _fold(scale, other):
        return scale if other is None else scale * other.reshape(-1)

    def _combine_vals(self, name: str, split: "_SplitSliced"):
        """Second-stage values of a split layout for this view's edge values (made once per view and layout)."""
        if self._coo_vals is None or not split.has_light:
            return None
        v = self._v.get(name + "/combine")
        if v is None:
            v = self._v[name + "/combine"] = split.combine_values(self._coo_vals)
        return v

    def _vals_for(self, layout: str, eid: torch.Tensor):
        if self._coo_vals is None:
            return None
        v = self._v.get(layout)
        if v is None:
            v = self._v[layout] = gather_f32(self._coo_vals, eid)
        return v

    # -- structure accessors (kept as attributes of the public surface) ----------------------------
    n_dst = property(lambda self: self._S.n_dst)
    n_src = property(lambda self: self._S.n_src)
    indptr = property(lambda self: self._S.indptr)
    indices = property(lambda self: self._S.indices)
    eid = property(lambda self: self._S.eid)
    plan = property(lambda self: self._S.plan)
    regular = property(lambda self: self._S.regular)
    regular_t = property(lambda self: self._S.regular_t)
    _sliced = property(lambda self: self._S.sliced)
    _sliced_t = property(lambda self: self._S.sliced_t)

    @property
    def vals(self):
        """Edge values in CSR order (None for an unweighted graph)."""
        return self._vals_for("csr", self._S.eid)

    @property
    def nnz(self) -> int:
        return int(self._S.indices.shape[0])

    @property
    def device(self):
        return self._S.indptr.device

    @staticmethod
    def _is_regular(max_deg: int, nnz: int, n_rows: int) -> bool:
        # the sliced kernel streams a (row, slice) segment sequentially in one lane group: rows up
        # to ~32x the average are a tail effect, rows of 10^5..10^6 edges (power laws) are not
        return max_deg <= max(1024, 32 * (nnz // max(n_rows, 1)))

    def _validate(self, flag: torch.Tensor):
        """One host sync: the range flag of the device COO->CSR and the maximum in-degree."""
        S = self._S
        if self.nnz == 0:
            S.regular = True
            return
        bad, max_deg = torch.stack([flag.reshape(()), (S.indptr[1:] - S.indptr[:-1]).max()]).tolist()
        if bad:  # error path only: say which side and by how much
            dlo, dhi, slo, shi = (int(v) for v in torch.stack([S.dst.min(), S.dst.max(), S.src.min(), S.src.max()]).tolist())
            if dlo < 0 or dhi >= S.n_dst:
                raise RuntimeError("destination id out of range [0, %d): min %d max %d" % (S.n_dst, dlo, dhi))
            raise RuntimeError("source id out of range [0, %d): min %d max %d" % (S.n_src, slo, shi))
        S.max_deg = int(max_deg)
        S.regular = self._is_regular(int(max_deg), self.nnz, S.n_dst)

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _decide_regular(self, X, transposed: bool):
        """Settle ``regular`` / ``regular_t`` of an unchecked build when a product large enough to care arrives."""
        S = self._S
        n_cols = S.n_dst if transposed else S.n_src
        small = X.dim() != 2 or n_cols * X.shape[1] * 4 < min(SLICED_MIN_TABLE_BYTES, SPLIT_MIN_TABLE_BYTES)
        # a VALIDATED build has synchronised once already and reads the transposed graph's longest row back whatever the
        # size (it also tells whether the planned form's second launch can be skipped: _plan_if_needed)
        if (small and not S.validated) or not X.is_cuda or self._capturing():
            return  # stays unknown (treated as "not regular" for this product): small tables never take the sliced form
        indptr = self._t_struct()[0] if transposed else S.indptr
        max_deg = int((indptr[1:] - indptr[:-1]).max()) if self.nnz else 0
        if transposed:
            S.max_deg_t, S.regular_t = max_deg, self._is_regular(max_deg, self.nnz, S.n_src)
        else:
            S.max_deg, S.regular = max_deg, self._is_regular(max_deg, self.nnz, S.n_dst)

    def _use_sliced(self, F: int, n_rows: int, n_cols: int, regular: bool) -> bool:
        if FORCE_KERNEL is not None:  # debugging / A-B aid: DGMI_FORCE_KERNEL=planned|sliced
            return FORCE_KERNEL == "sliced" and F % 4 == 0 and n_rows * SlicedCSR.N_SLICES < 2 ** 31 - 1
        table = n_cols * F * 4
        if not (bool(regular) and F % 4 == 0 and n_rows * SlicedCSR.N_SLICES < 2 ** 31 - 1) or table < SLICED_MIN_TABLE_BYTES:
            return False
        degree = self.nnz / max(n_rows, 1)
        if table <= _table_cap(SLICED_TIERS, degree):
            return True
        lo_deg, lo, hi = SLICED_LOW_DEGREE
        return degree >= lo_deg and lo <= table <= hi

    def _use_split(self, F: int, n_rows: int, n_cols: int, regular) -> bool:
        """Long-row graphs (power laws): virtual rows for the heavy rows, the light rows through the second stage.
        Only for graphs that were validated at build (the split needs host readbacks).  Rule refitted in round 4
        (tools/kernel_choice_sweep.py, Zipf(1.1) degrees, 6.4 M edges: planned / split time 1.15-2.7 on every table up
        to 275 MB at EVERY average degree from 16 — the light-row second stage made the split form independent of the
        degree —, 1.0-1.04 at 410-550 MB, 0.7-0.9 from 800 MB unless the average degree is in the hundreds)."""
        if FORCE_KERNEL is not None or regular is not False or not self._S.validated:
            return False
        table = n_cols * F * 4
        n_virtual_bound = n_rows + self.nnz // SPLIT_ROW_EDGES
        if F % 4 != 0 or n_virtual_bound * SlicedCSR.N_SLICES >= 2 ** 31 - 1 or table < SPLIT_MIN_TABLE_BYTES:
            return False
        degree = self.nnz / max(n_rows, 1)
        if degree < SPLIT_MIN_DEGREE:
            return False
        return table <= (SPLIT_MAX_TABLE_BYTES_DENSE if degree >= SPLIT_DENSE_DEGREE else SPLIT_MAX_TABLE_BYTES)

    def _few_long_rows(self, F: int, n_rows: int, n_cols: int) -> bool:
        """A REGULAR graph of few, long rows whose XCD slices of the table exceed an L2 (a config-5 edge-scaled shard:
        6 250 rows of 1 600 edges over 100 000 sources): below 32 768 rows the launcher keeps one full-width pass (half-
        width groups would leave too few waves), so the slice spills — cut into 256-edge virtual rows there are enough rows
        for the column passes again (tools/long_rows_probe.py: 0.371 -> 0.322 ms, 0.345 -> 0.302, 0.339 -> 0.308; no gain
        where a slice already fits: 0.247 -> 0.254)."""
        if FORCE_KERNEL is not None or not self._S.validated or n_rows >= COLUMN_PASS_MIN_ROWS:
            return False
        slice_bytes = (n_cols + SlicedCSR.N_SLICES - 1) // SlicedCSR.N_SLICES * F * 4
        return self.nnz >= LONG_ROW_MIN_DEGREE * max(n_rows, 1) and slice_bytes > (4 << 20)

    @staticmethod
    def _plan_if_needed(plan, max_deg):
        """A plan only cuts rows longer than its chunk; when a readback has told that none is (every kNN-4 graph, every
        small slice), the wave-per-row launch does the same work in ONE launch — the planned form's second kernel,
        which adds the chunks of long rows, would start, find none and leave: 4-5 us each, 16 of them per training step
        at the reference's dataset sizes."""
        if plan is not None and max_deg is not None and max_deg <= plan.chunk:
            return None
        return plan

    def _t_struct(self):
        S = self._S
        if S.t is None:
            indptr_t, indices_t, eid_t = csr_from_coo(S.src, S.dst, S.n_src)
            S.t = (indptr_t, indices_t, eid_t, build_plan(indptr_t, self.nnz) if S.planned else None)
        return S.t

    def transposed(self):
        """(indptr_t, indices_t, vals_t, plan_t): CSR of the reversed edges, rows = source nodes."""
        indptr_t, indices_t, eid_t, plan_t = self._t_struct()
        return indptr_t, indices_t, self._vals_for("csr_t", eid_t), plan_t

    def _run(self, indptr, indices, vals, plan, n_rows, n_cols, X, col_scale, row_scale, out, eid=None, epi=None):
        dev = indptr.device
        if not X.is_cuda or X.device != dev:
            _require_device(indptr, X)
        if X.dim() == 2 and X.shape[0] != n_cols:
            raise RuntimeError("X has %d rows, the graph has %d source nodes" % (X.shape[0], n_cols))
        return _launch_spmm(dev, indptr, indices, vals, X, None if col_scale is None else col_scale.reshape(-1),
                            None if row_scale is None else row_scale.reshape(-1), out, plan, n_rows, n_cols, 0, 0,
                            eid if self._keep is not None else None, self._keep, epi)

    def spmm(self, X, src_scale=None, dst_scale=None, out=None, epi=None):
        """``diag(dst_scale) A diag(src_scale) X`` (no autograd).  Picks the XCD-local sliced kernel when the
        feature table is a few L2s large and the graph is regular, else the planned kernel.  ``epi``: output epilogue
        (act, slope, out_mask, mask_scale), fused into the kernel that writes the result."""
        S = self._S
        if S.regular is None:
            self._decide_regular(X, False)
        long_rows = X.dim() == 2 and bool(S.regular) and self._few_long_rows(X.shape[1], S.n_dst, S.n_src)
        if (X.dim() == 2 and not long_rows and self._use_sliced(X.shape[1], S.n_dst, S.n_src, S.regular)
                and _sliced_ok(X, out)):
            if S.sliced is None:
                S.sliced = SlicedCSR.from_csr(S.indptr, S.indices, S.eid, S.n_dst, S.n_src)  # one partition pass, no sort
            mult = self._mult_ids("sliced", S.sliced)
            if mult is not None:  # values = scale x multiplicity: a destination (D^-1 M) or a source (M D^-1) scale
                if mult[0] == "row":
                    dst_scale = self._fold(mult[1], dst_scale)
                else:
                    src_scale = self._fold(mult[1], src_scale)
            c = self._compacted("sliced", S.sliced, mult)
            if c is not None:
                return c.spmm(X, src_scale, dst_scale, out, epi=epi)
            if mult is not None:
                return S.sliced.spmm(X, src_scale, dst_scale, out, vals=None, keep=self._keep, epi=epi, indices=mult[2], id_mult=True)
            return S.sliced.spmm(X, src_scale, dst_scale, out, vals=self._vals_for("sliced", S.sliced.eid), keep=self._keep,
                                 epi=epi)
        if X.dim() == 2 and _sliced_ok(X, out) and (self._use_split(X.shape[1], S.n_dst, S.n_src, S.regular) or (
                long_rows and self._use_sliced(X.shape[1], S.n_dst, S.n_src, S.regular))):
            if S.split is None:
                S.split = _SplitSliced(S.indptr, S.eid, S.src, S.n_dst, S.n_src)
            c = self._compacted("split", S.split.sliced)
            return S.split.spmm(X, _prep_scale(src_scale, S.n_src, "src_scale"), _prep_scale(dst_scale, S.n_dst, "dst_scale"),
                                out, None if c is not None else self._vals_for("split", S.split.sliced.eid), keep=self._keep,
                                epi=epi, compacted=c, c_vals=self._combine_vals("split", S.split))
        return self._run(S.indptr, S.indices, self.vals, self._plan_if_needed(S.plan, S.max_deg), S.n_dst, S.n_src, X,
                         src_scale, dst_scale, out, S.eid, epi)

    def spmm_t(self, dY, src_scale=None, dst_scale=None, out=None):
        """``diag(src_scale) A^T diag(dst_scale) dY`` — the backward of :meth:`spmm`."""
        S = self._S
        indptr_t, indices_t, eid_t, plan_t = self._t_struct()
        if S.regular_t is None:  # one-time readback of the reversed graph's maximum degree, when it can matter
            self._decide_regular(dY, True)
        long_rows = dY.dim() == 2 and bool(S.regular_t) and self._few_long_rows(dY.shape[1], S.n_src, S.n_dst)
        if (dY.dim() == 2 and not long_rows and self._use_sliced(dY.shape[1], S.n_src, S.n_dst, S.regular_t)
                and _sliced_ok(dY, out)):
            if S.sliced_t is None:
                S.sliced_t = SlicedCSR.from_csr(indptr_t, indices_t, eid_t, S.n_src, S.n_dst)
            mult = self._mult_ids("sliced_t", S.sliced_t)
            if mult is not None:  # (D^-1 M)^T = M^T D^-1: the scale multiplies the gathered (destination-node) rows; (M D^-1)^T: the output rows
                if mult[0] == "row":
                    dst_scale = self._fold(mult[1], dst_scale)
                else:
                    src_scale = self._fold(mult[1], src_scale)
            c = self._compacted("sliced_t", S.sliced_t, mult)
            if c is not None:
                return c.spmm(dY, dst_scale, src_scale, out)
            if mult is not None:
                return S.sliced_t.spmm(dY, dst_scale, src_scale, out, vals=None, keep=self._keep, indices=mult[2], id_mult=True)
            return S.sliced_t.spmm(dY, dst_scale, src_scale, out, vals=self._vals_for("sliced_t", S.sliced_t.eid),
                                   keep=self._keep)
        if dY.dim() == 2 and _sliced_ok(dY, out) and (self._use_split(dY.shape[1], S.n_src, S.n_dst, S.regular_t) or (
                long_rows and self._use_sliced(dY.shape[1], S.n_src, S.n_dst, S.regular_t))):
            if S.split_t is None:
                S.split_t = _SplitSliced(indptr_t, eid_t, S.dst, S.n_src, S.n_dst)
            c = self._compacted("split_t", S.split_t.sliced)
            return S.split_t.spmm(dY, _prep_scale(dst_scale, S.n_dst, "dst_scale"), _prep_scale(src_scale, S.n_src, "src_scale"),
                                  out, None if c is not None else self._vals_for("split_t", S.split_t.sliced.eid), keep=self._keep,
                                  compacted=c, c_vals=self._combine_vals("split_t", S.split_t))
        return self._run(indptr_t, indices_t, self._vals_for("csr_t", eid_t), self._plan_if_needed(plan_t, S.max_deg_t), S.n_src, S.n_dst, dY,
                         dst_scale, src_scale, out, eid_t)


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, g: CSRGraph, src_scale, dst_scale):
        ctx.g = g
        ctx.save_for_backward(src_scale, dst_scale)
        return g.spmm(X, src_scale, dst_scale)

    @staticmethod
    def backward(ctx, dY):
        src_scale, dst_scale = ctx.saved_tensors
        dX = None
        if ctx.needs_input_grad[0]:
            # dX = diag(src_scale) A^T diag(dst_scale) dY : the same kernel on the reversed
            # edges with the two scales swapped.  (An expanded / strided gradient — `y.sum().backward()` hands one over —
            # is made dense first: the XCD-local form needs 16-B aligned rows, and a 51 MB copy is 20 us against the 0.4 ms
            # the planned kernel would lose.)
            if not dY.is_contiguous():
                dY = dY.contiguous()
            dX = ctx.g.spmm_t(dY, src_scale, dst_scale)
        return dX, None, None, None


class _SpMMEpilogue(torch.autograd.Function):
    """``mask * mask_scale * act(diag(ds) A diag(ss) X)`` with the activation and the dropout mask
    applied by the kernel that writes the product (f3, reference layers.py:134-138)."""

    @staticmethod
    def forward(ctx, X, g: CSRGraph, src_scale, dst_scale, act, slope, mask, mask_scale):
        y = g.spmm(X, src_scale, dst_scale, epi=(act, slope, mask, mask_scale))
        ctx.g, ctx.epi = g, (act, slope, mask_scale)
        ctx.save_for_backward(src_scale, dst_scale, y, mask)
        return y

    @staticmethod
    def backward(ctx, dY):
        src_scale, dst_scale, y, mask = ctx.saved_tensors
        act, slope, mask_scale = ctx.epi
        dX = None
        if ctx.needs_input_grad[0]:
            g_pre = epilogue_backward(dY.contiguous(), y, mask, act, slope, mask_scale)
            dX = ctx.g.spmm_t(g_pre, src_scale, dst_scale)
        return dX, None, None, None, None, None, None, None


def colsum_rows_(feat_ext, coef, n: int, R: int, i0: int):
    """In place: rows ``[n*R, n*R + B)`` of ``feat_ext`` <- ``coef @ feat_ext[i0::R][:n]`` — the per-block column sums of
    the complement form (``dgmi_weighted_colsum_f32``)."""
    _require_device(feat_ext, coef)
    _T.colsum_rows_(feat_ext, coef, n, R, i0)
    return feat_ext


def colsum_rows_backward_(gf, coef, gs, n: int, R: int, i0: int):
    """In place: ``gf[i0::R] += coef^T @ gs`` (``dgmi_rank_add_f32``)."""
    _require_device(gf, coef, gs)
    _T.colsum_rows_backward_(gf, coef, gs, n, R, i0)
    return gf


def epilogue_backward(dY, Y, mask, act, slope, mask_scale):
    """``dY * act'(Y) * mask * mask_scale`` in one pass (``dgmi_epilogue_backward_f32``)."""
    _require_device(dY, Y, mask)
    return _T.epilogue_backward(dY, Y, mask, act, slope, mask_scale)


def spmm_csr_act_dropout(g: CSRGraph, X, src_scale=None, dst_scale=None, act: int = 0, slope: float = 0.0,
                         mask: Optional[torch.Tensor] = None, mask_scale: float = 1.0) -> torch.Tensor:
    """:func:`spmm_csr` followed by ``leaky_relu`` (``act=1``) and a dropout keep ``mask`` (0/1 floats of
    the output's shape, scaled by ``mask_scale``), both inside the product's last kernel."""
    if src_scale is not None and src_scale.requires_grad or dst_scale is not None and dst_scale.requires_grad:
        raise RuntimeError("spmm_csr: gradients w.r.t. the diagonal scales are not part of the path")
    return _SpMMEpilogue.apply(X, g, src_scale, dst_scale, act, slope, mask, mask_scale)


def spmm_csr(g: CSRGraph, X: torch.Tensor, src_scale: Optional[torch.Tensor] = None,
             dst_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Y = diag(dst_scale) · A · diag(src_scale) · X`` with autograd w.r.t. ``X`` only.

    The reference needs no other gradient: adjacency values are constants (utils.py:24-27,
    augmentation.py:124) and ``ci``/``cj`` are non-learnable node data (data_loader.py:487-488).
    """
    if src_scale is not None and src_scale.requires_grad or dst_scale is not None and dst_scale.requires_grad:
        raise RuntimeError("spmm_csr: gradients w.r.t. the diagonal scales are not part of the path")
    return _SpMM.apply(X, g, src_scale, dst_scale)


# ---------------------------------------------------------------------------------------------
# (f2) decoder edge gather-concat: graph.apply_edges(udf_u_mul_e) — layers.py:364,378-379
# ---------------------------------------------------------------------------------------------
def _check_tables(A, B, n_src, n_dst):
    """The feature tables must cover the id ranges the edge list was validated against
    (``EdgePairs``): DGL would raise on a node-data / node-count mismatch; reading past a short
    table on the GPU must not be an option."""
    if n_src is not None and A.shape[0] < n_src:
        raise RuntimeError("source feature table has %d rows, the edge list addresses %d source nodes" % (A.shape[0], n_src))
    if n_dst is not None and B.shape[0] < n_dst:
        raise RuntimeError("destination feature table has %d rows, the edge list addresses %d destination nodes"
                           % (B.shape[0], n_dst))


def gather_concat_raw(src, dst, A, B, out=None, n_src=None, n_dst=None) -> torch.Tensor:
    """``out[e] = cat(A[src[e]], B[dst[e]])`` through ``dgmi_gather_concat_f32`` (no autograd).
    ``n_src`` / ``n_dst``: the node counts ``src`` / ``dst`` ids were range-checked against."""
    _require_device(src, dst, A, B, out)
    _check_tables(A, B, n_src, n_dst)
    y = _T.gather_concat_raw(src, dst, A, B)
    if out is not None:
        out.copy_(y)
        return out
    return y


class EdgePairs:
    """The decoder graph's edge list (data_loader.py:492-509) in the layout the kernels read:
    int32 ``src``/``dst`` ids plus, built lazily for the backward, the two CSRs whose
    ``indices`` are *edge ids* grouped by source node and by destination node (so that
    ``copy_e -> sum`` is the SpMM kernel gathering rows of d_out)."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, n_src: int, n_dst: int, check_range: bool = True):
        _require_device(src, dst)
        self.src = src.to(torch.int32).contiguous()
        self.dst = dst.to(torch.int32).contiguous()
        self.n_src, self.n_dst = int(n_src), int(n_dst)
        self.E = int(self.src.shape[0])
        if check_range and self.E:
            lo = torch.stack([self.src.min(), self.dst.min()]).min()
            hi_s, hi_d = self.src.max(), self.dst.max()
            lo, hi_s, hi_d = (int(v) for v in torch.stack([lo, hi_s, hi_d]).tolist())
            if lo < 0 or hi_s >= self.n_src or hi_d >= self.n_dst:
                raise RuntimeError("decoder edge id out of range")
        self._by_src = self._by_dst = None

    def _group(self, key, n):
        iota = torch.arange(self.E, dtype=torch.int32, device=key.device)
        # every row of the gathered (E, F) matrix is used exactly ONCE (the column ids are a permutation of the edge
        # ids): no reuse for the XCD-local form to keep in L2 — vouch "not regular" so that it is never chosen
        g = CSRGraph(key, iota, n, self.E, check_range=False, regular=False, regular_t=False)
        return g

    def by_src(self) -> "CSRGraph":
        if self._by_src is None:
            self._by_src = self._group(self.src, self.n_src)
        return self._by_src

    def by_dst(self) -> "CSRGraph":
        if self._by_dst is None:
            self._by_dst = self._group(self.dst, self.n_dst)
        return self._by_dst


class _GatherConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, pairs: EdgePairs):
        ctx.pairs = pairs
        ctx.Fa, ctx.Fb = A.shape[1], B.shape[1]
        return gather_concat_raw(pairs.src, pairs.dst, A, B, n_src=pairs.n_src, n_dst=pairs.n_dst)

    @staticmethod
    def backward(ctx, dOut):
        pairs, Fa, Fb = ctx.pairs, ctx.Fa, ctx.Fb
        dOut = dOut.contiguous()
        dA = dB = None
        if ctx.needs_input_grad[0]:  # dA[u] = sum over edges leaving u of d_out[e, :Fa]
            dA = pairs.by_src().spmm(dOut[:, :Fa])
        if ctx.needs_input_grad[1]:  # dB[v] = sum over edges entering v of d_out[e, Fa:]
            dB = pairs.by_dst().spmm(dOut[:, Fa:])
        return dA, dB, None


def gather_concat(pairs: EdgePairs, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Differentiable ``cat(A[src], B[dst])`` over the decoder edges."""
    return _GatherConcat.apply(A, B, pairs)


def gather_add_raw(src, dst, A, B, bias=None, out=None, n_src=None, n_dst=None, act: int = 0) -> torch.Tensor:
    """``out[e] = A[src[e]] + B[dst[e]] (+ bias)`` through ``dgmi_gather_add_f32`` (no autograd); ``act=1``: relu of it,
    in the same pass."""
    _require_device(src, dst, A, B, bias, out)
    _check_tables(A, B, n_src, n_dst)
    y = _T.gather_add_raw(src, dst, A, B, bias, int(act))
    if out is not None:
        out.copy_(y)
        return out
    return y


class _GatherAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, bias, pairs: EdgePairs):
        ctx.pairs = pairs
        return gather_add_raw(pairs.src, pairs.dst, A, B, bias, n_src=pairs.n_src, n_dst=pairs.n_dst)

    @staticmethod
    def backward(ctx, dOut):
        pairs = ctx.pairs
        dOut = dOut.contiguous()
        dA = pairs.by_src().spmm(dOut) if ctx.needs_input_grad[0] else None
        dB = pairs.by_dst().spmm(dOut) if ctx.needs_input_grad[1] else None
        dbias = dOut.sum(0) if ctx.needs_input_grad[2] else None
        return dA, dB, dbias, None


def gather_add(pairs: EdgePairs, A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """Differentiable ``A[src] + B[dst] (+ bias)`` over the decoder edges."""
    return _GatherAdd.apply(A, B, bias, pairs)


class _GatherAddReluDropout(torch.autograd.Function):
    """``dropout(relu(A[src] + B[dst] + bias))`` — the decoder's first stage (layers.py:364-367 with ``lin1`` folded
    into the node tables).  Forward: the relu rides in the gather-add kernel's store (one E x F pass less), the dropout
    is torch's own (same RNG consumption as ``nn.Dropout`` on that shape).  Backward: ONE gating pass from the output
    (``y > 0`` exactly where the sum was positive and the element was kept: ``dgmi_epilogue_backward_f32`` act 2), then
    the two segment sums of :class:`_GatherAdd`."""

    @staticmethod
    def forward(ctx, A, B, bias, pairs: EdgePairs, p: float):
        y = gather_add_raw(pairs.src, pairs.dst, A, B, bias, n_src=pairs.n_src, n_dst=pairs.n_dst, act=1)
        y = torch.nn.functional.dropout(y, p, True)
        ctx.pairs, ctx.scale = pairs, 1.0 / (1.0 - p)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dY):
        (y,) = ctx.saved_tensors
        pairs = ctx.pairs
        dZ = epilogue_backward(dY.contiguous(), y, None, 2, 0.0, ctx.scale)
        dA = pairs.by_src().spmm(dZ) if ctx.needs_input_grad[0] else None
        dB = pairs.by_dst().spmm(dZ) if ctx.needs_input_grad[1] else None
        dbias = dZ.sum(0) if ctx.needs_input_grad[2] else None
        return dA, dB, dbias, None, None


def gather_add_relu_dropout(pairs: EdgePairs, A, B, bias, p: float):
    """Differentiable ``dropout(relu(A[src] + B[dst] + bias), p)`` in training mode (``0 < p < 1``)."""
    return _GatherAddReluDropout.apply(A, B, bias, pairs, float(p))


# ---------------------------------------------------------------------------------------------
# (f4) cosine-similarity kNN — data_loader.py:312-344 without the N x N similarity matrix
# ---------------------------------------------------------------------------------------------
def knn_cosine_supported(N: int, D: int, k: int) -> bool:
    """Whether the fused MFMA kNN kernel takes this shape (``dgmi_knn_cosine_supported``)."""
    return bool(_L.dgmi_knn_cosine_supported(int(N), int(D), int(k)))


def knn_cosine_topk(xn: torch.Tensor, k: int) -> torch.Tensor:
    """``(N, k)`` int32 ids of the k largest cosine similarities per row of the row-normalised ``xn``
    (self included, descending) — ``dgmi_knn_cosine_topk_f32``: fp32-MFMA similarity tiles reduced to
    a running top-k on chip."""
    _require_device(xn)
    return _T.knn_cosine_topk(xn, int(k))


# ---------------------------------------------------------------------------------------------
# (D3) edge-dropout selection — augmentation.py:48-52, 114-118
# ---------------------------------------------------------------------------------------------
def random_subset_select(E: int, keep: int, seed: int, device, e_offset: int = 0) -> torch.Tensor:
    """The 8-word description (int32 tensor on ``device``) of a uniformly random subset of exactly
    ``keep`` of ``E`` edges, a deterministic function of ``(seed, E, keep)``
    (``dgmi_random_subset_select``): what ``CSRGraph.dropped`` and the kernels consume.  Nothing of
    size E is written and nothing synchronises."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dream_gnn_amd ops run on the MI355X only: got device %s" % device)
    return _T.random_subset_select(torch.empty(0, device=device), E, keep, _signed64(seed), e_offset)


def _signed64(seed: int) -> int:
    seed &= 0xFFFFFFFFFFFFFFFF
    return seed - (1 << 64) if seed >= 1 << 63 else seed


def random_subset_select_batch(Es, keeps, seeds, device, e_offsets=None) -> torch.Tensor:
    """``random_subset_select`` for several edge lists at once: ``(n, 8)`` descriptions from ONE series
    of launches per 8 lists (``dgmi_random_subset_select_batch``) — the training step drops edges on
    4 relations and 4 similarity graphs together (train.py:267), and at dataset scale the launches
    are the cost of the selection."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dream_gnn_amd ops run on the MI355X only: got device %s" % device)
    n = len(Es)
    offs = [0] * n if e_offsets is None else list(e_offsets)
    if isinstance(seeds, torch.Tensor):
        # seeds in device memory (int64, one per list), read by the kernels (``dgmi_random_subset_select_batch_dseed``):
        # no host value goes into the launches, so the call can sit inside a captured HIP graph and still select new
        # subsets on every replay
        _require_device(seeds)
        seeds = seeds.reshape(-1).to(torch.int64).contiguous()
        parts = [_T.random_subset_select_batch_dseed(seeds[i:i + 8], list(Es[i:i + 8]), list(keeps[i:i + 8]), offs[i:i + 8])
                 for i in range(0, n, 8)]
        return parts[0] if len(parts) == 1 else torch.cat(parts)
    like = torch.empty(0, device=device)
    parts = [_T.random_subset_select_batch(like, list(Es[i:i + 8]), list(keeps[i:i + 8]),
                                           [_signed64(s) for s in seeds[i:i + 8]], offs[i:i + 8])
             for i in range(0, n, 8)]
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def keep_mask(desc: torch.Tensor, E: int) -> torch.Tensor:
    """float 0/1 mask over edges [0, E) under the subset description(s) ``desc`` (``dgmi_keep_mask_f32``)."""
    _require_device(desc)
    return _T.keep_mask(_prep_keep(desc), E)


def random_subset_mask(E: int, keep: int, seed: int, device) -> torch.Tensor:
    """float 0/1 mask over E edges with exactly ``keep`` ones, a uniformly random subset that is a
    deterministic function of ``(seed, E, keep)`` (``dgmi_random_subset_mask_f32``)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dream_gnn_amd ops run on the MI355X only: got device %s" % device)
    mask = torch.empty(E, dtype=torch.float32, device=device)
    nbytes = int(_L.dgmi_random_subset_workspace_bytes())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    with _guard(device):
        _lib.check(_L.dgmi_random_subset_mask_f32(E, keep, seed & 0xFFFFFFFFFFFFFFFF, _ptr(mask), _ptr(ws), nbytes,
                                                  _stream(device)), "dgmi_random_subset_mask_f32")
    return mask
