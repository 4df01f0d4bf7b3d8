"""Light graph containers consumed by the drop-in modules in place of DGL graphs.

The reference hands its layers a DGL heterograph built by
``DrugDataLoader._generate_enc_graph`` (reference data_loader.py:400-490): canonical edge
types ``('drug', r, 'disease')`` and ``('disease', 'rev-'+r, 'drug')`` for each rating ``r``,
with per-node-type data ``ci`` / ``cj`` (N,1) fp32.  ``HeteroGraph`` keeps that shape of API —
only the members the reference's layers / augmentation actually touch — on top of device
COO tensors and lazily built device CSRs (``ops.CSRGraph``).  Nothing here depends on DGL.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, Optional, Tuple

import torch

from . import ops

# (f3) complement form of a near-complete relation (HeteroGraph.fused_relations_complement): taken from this
# share of the cells up (a value > 1 switches it off); the dense presence map it is built from stays small
COMPLEMENT_MIN_DENSITY = float(os.environ.get("DGMI_COMPLEMENT_MIN_DENSITY", "0.5"))
COMPLEMENT_MAX_CELLS = 1 << 26

CanonicalEType = Tuple[str, str, str]


class _NodeView:
    def __init__(self, data: dict):
        self.data = data


class _NodeSpace:
    """``graph.nodes[ntype].data`` (layers.py:362-363, data_loader.py:487-488)."""

    def __init__(self, store: Dict[str, dict]):
        self._store = store

    def __getitem__(self, ntype: str) -> _NodeView:
        return _NodeView(self._store[ntype])


class RelationGraph:
    """One canonical edge type of a :class:`HeteroGraph`: edges ``src_type -> dst_type``.

    This is what ``GCMCGraphConv.forward`` receives (layers.py:169): it exposes ``srcdata`` /
    ``dstdata`` (views of the parent's node data), ``number_of_src_nodes`` and ``local_scope``
    (layers.py:174-190), plus ``csr`` — the destination-major CSR the HIP kernel reads.
    """

    def __init__(self, canonical: CanonicalEType, src: torch.Tensor, dst: torch.Tensor,
                 n_src: int, n_dst: int, srcdata: dict, dstdata: dict):
        self.canonical = canonical
        self.src, self.dst = src, dst
        self.n_src, self.n_dst = int(n_src), int(n_dst)
        self.srcdata, self.dstdata = srcdata, dstdata
        self._csr: Optional[ops.CSRGraph] = None
        self.trusted = False  # ids already validated (edge lists derived from a checked graph)
        self.regular_hint = None  # (regular, regular_t) inherited from the parent graph (edge dropout)

    # -- the DGL surface the reference's layer code uses ---------------------------------
    def number_of_src_nodes(self) -> int:
        return self.n_src

    def number_of_dst_nodes(self) -> int:
        return self.n_dst

    def number_of_edges(self) -> int:
        return int(self.src.shape[0])

    def edges(self):
        return self.src, self.dst

    @property
    def device(self):
        return self.src.device

    def in_degrees(self) -> torch.Tensor:
        return torch.bincount(self.dst.long(), minlength=self.n_dst)

    def out_degrees(self) -> torch.Tensor:
        return torch.bincount(self.src.long(), minlength=self.n_src)

    @contextlib.contextmanager
    def local_scope(self):
        """Node-data keys written inside the scope vanish on exit (layers.py:174)."""
        saved_src, saved_dst = dict(self.srcdata), dict(self.dstdata)
        try:
            yield self
        finally:
            for live, saved in ((self.srcdata, saved_src), (self.dstdata, saved_dst)):
                live.clear()
                live.update(saved)

    # -- kernel-side layout -----------------------------------------------------------------
    @property
    def csr(self) -> ops.CSRGraph:
        if self._csr is None:
            hint, hint_t = self.regular_hint if self.regular_hint is not None else (None, None)
            self._csr = ops.CSRGraph(self.dst, self.src, self.n_dst, self.n_src, check_range=not self.trusted,
                                     regular=hint, regular_t=hint_t)
        return self._csr


class DroppedRelation(RelationGraph):
    """A relation after edge dropout (augmentation.py:13-89), kept as *parent relation + which
    edges survive* instead of a re-sorted copy.

    ``csr`` is a view of the parent's CSR: no sort, no plan rebuild, no host sync — the
    per-iteration graph churn of train.py:267 becomes one subset selection.  With the default
    ``"select"`` selection the subset is an 8-word description (``ops.random_subset_select``) that
    the kernels evaluate per edge while they walk the parent's layouts (``CSRGraph.dropped``):
    nothing of size E is written, dropped edges are skipped.  With ``"randperm"`` (the reference's
    literal procedure) it is a 0/1 value mask over the parent CSR (dropped edges contribute
    ``0 * x``).  The kept edge list (``src`` / ``dst``) is materialised only if somebody asks for
    it (in permutation order when the subset came from ``randperm``, as the reference builds it;
    in ascending order otherwise).
    """

    def __init__(self, parent: RelationGraph, n_keep: int, srcdata: dict, dstdata: dict,
                 keep_idx: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                 desc: Optional[torch.Tensor] = None):
        self.canonical = parent.canonical
        self.parent = parent
        self.n_keep = int(n_keep)
        self._keep_idx, self._mask, self.desc = keep_idx, mask, desc
        self.n_src, self.n_dst = parent.n_src, parent.n_dst
        self.srcdata, self.dstdata = srcdata, dstdata
        self._csr = None
        self._pairs = None
        self._lists = None
        self.trusted = True
        self.regular_hint = None

    @property
    def keep_idx(self) -> torch.Tensor:
        """Positions of the kept edges in the parent's edge list."""
        if self._keep_idx is None:
            self._keep_idx = torch.nonzero(self.keep_mask(), as_tuple=True)[0]  # host sync: only on demand
        return self._keep_idx

    def keep_mask(self) -> torch.Tensor:
        """float 0/1 over the parent's edges (materialised on demand)."""
        if self._mask is None:
            if self.desc is not None:
                self._mask = ops.keep_mask(self.desc, self.parent.number_of_edges())
            else:
                m = torch.zeros(self.parent.number_of_edges(), dtype=torch.float32, device=self._keep_idx.device)
                self._mask = m.index_fill_(0, self._keep_idx, 1.0)  # (m[idx] = 1.0 would synchronise with the host)
        return self._mask

    def _materialise(self):
        if self._lists is None:
            self._lists = (self.parent.src[self.keep_idx], self.parent.dst[self.keep_idx])
        return self._lists

    src = property(lambda self: self._materialise()[0], lambda self, v: self._set(0, v))
    dst = property(lambda self: self._materialise()[1], lambda self, v: self._set(1, v))

    def _set(self, i, v):
        lists = list(self._materialise())
        lists[i] = v
        self._lists = tuple(lists)

    def number_of_edges(self) -> int:
        return self.n_keep

    @property
    def device(self):
        return self.parent.device

    @property
    def csr(self) -> ops.CSRGraph:
        if self._csr is None:
            base = self.parent.csr
            self._csr = base.dropped(self.desc) if self.desc is not None else base.masked(self.keep_mask())
        return self._csr


class HeteroGraph:
    """Bipartite heterograph: ``{(src_type, etype, dst_type): (src_ids, dst_ids)}``.

    Constructor arguments follow ``dgl.heterograph(data_dict, num_nodes_dict=...)`` as the
    reference calls it (data_loader.py:448, augmentation.py:65).  Edge order inside an edge
    type is the caller's order; the CSR build is stable, so that order is the in-row
    summation order, as with DGL.  Node types and relation triples are kept SORTED, as
    ``dgl.heterograph`` does (dgl/convert.py sorts both "to have a deterministic order" — from
    memory of DGL's public source; DGL is absent here, so this is unpinned): it is the order
    ``HeteroGraphConv`` runs the per-relation modules — i.e. the order of their dropout draws —
    and the order ``random_edge_dropout`` walks the edge types (augmentation.py:24).
    """

    def __init__(self, data_dict: Dict[CanonicalEType, Tuple[torch.Tensor, torch.Tensor]],
                 num_nodes_dict: Dict[str, int], device=None):
        self._num_nodes = {k: int(num_nodes_dict[k]) for k in sorted(num_nodes_dict)}
        self._ndata: Dict[str, dict] = {nt: {} for nt in self._num_nodes}
        self._rels: Dict[CanonicalEType, RelationGraph] = {}
        self.nodes = _NodeSpace(self._ndata)
        for can in sorted(data_dict):
            src, dst = data_dict[can]
            st, _, dt = can
            src = torch.as_tensor(src)
            dst = torch.as_tensor(dst)
            if device is not None:
                src, dst = src.to(device), dst.to(device)
            if src.shape != dst.shape or src.dim() != 1:
                raise ValueError("edge lists of %s must be two 1-D tensors of equal length" % (can,))
            self._rels[can] = RelationGraph(can, src, dst, self._num_nodes[st], self._num_nodes[dt],
                                            self._ndata[st], self._ndata[dt])

    # -- DGL-shaped accessors -----------------------------------------------------------------
    @property
    def ntypes(self):
        return list(self._num_nodes)

    @property
    def canonical_etypes(self):
        return list(self._rels)

    @property
    def etypes(self):
        return [c[1] for c in self._rels]

    @property
    def device(self):
        for r in self._rels.values():
            return r.device
        return torch.device("cpu")

    def _canonical(self, key) -> CanonicalEType:
        if isinstance(key, tuple):
            return key
        hits = [c for c in self._rels if c[1] == key]
        if len(hits) != 1:
            raise KeyError("edge type %r matches %d relations" % (key, len(hits)))
        return hits[0]

    def __getitem__(self, key) -> RelationGraph:
        return self._rels[self._canonical(key)]

    def number_of_nodes(self, ntype: str) -> int:
        return self._num_nodes[ntype]

    def number_of_edges(self, etype=None) -> int:
        if etype is None:
            return sum(r.number_of_edges() for r in self._rels.values())
        return self[etype].number_of_edges()

    def edges(self, etype=None):
        return self[etype].edges()

    def int(self) -> "HeteroGraph":
        """``graph.int()`` (train.py:199, evaluation.py:33): ids to int32, in place of a copy."""
        for r in self._rels.values():
            if r.src.dtype != torch.int32:
                r.src, r.dst = r.src.to(torch.int32), r.dst.to(torch.int32)
                r._csr = None
                r._pairs = None
                self.__dict__.pop("_fused", None)
        return self

    def to(self, device) -> "HeteroGraph":
        device = torch.device(device) if not isinstance(device, torch.device) else device
        for r in self._rels.values():
            if r.src.device != device:
                r.src, r.dst = r.src.to(device), r.dst.to(device)
                r._csr = None
                r._pairs = None
                self.__dict__.pop("_fused", None)
        for store in self._ndata.values():
            for k, v in list(store.items()):
                store[k] = v.to(device)
        return self

    # -- decoder-side surface (layers.py:360-365) ------------------------------------------------
    @contextlib.contextmanager
    def local_scope(self):
        saved = {nt: dict(d) for nt, d in self._ndata.items()}
        saved_e = getattr(self, "edata", None)
        self.edata = {}
        try:
            yield self
        finally:
            for nt, d in self._ndata.items():
                d.clear()
                d.update(saved[nt])
            if saved_e is None:
                del self.edata
            else:
                self.edata = saved_e

    def apply_edges(self, udf, etype=None):
        """``graph.apply_edges(udf)`` for a single-relation graph (the decoder graph,
        data_loader.py:508): the UDF sees per-edge source / destination node data."""
        rel = self[etype] if etype is not None else self._single()

        class _Edges:
            pass

        e = _Edges()
        e.src = {k: v.index_select(0, rel.src.long()) for k, v in rel.srcdata.items()}
        e.dst = {k: v.index_select(0, rel.dst.long()) for k, v in rel.dstdata.items()}
        if not hasattr(self, "edata"):
            self.edata = {}
        self.edata.update(udf(e))

    def fused_relations(self, dst_type: str):
        """All relations that end in ``dst_type`` as ONE destination-major CSR (f3).

        With R relations r = 0..R-1 (canonical order) from the same source type, edge (u -> v)
        of relation r becomes column ``R*u + r``: the row sum over the fused CSR is then the
        hetero ``'sum'`` aggregate of layers.py:98,129 when the source features are laid out
        ``[u][r][:]`` (one GEMM ``X @ [W_0 | ... | W_{R-1}]``).  In-row order: relation 0's
        edges in their input order, then relation 1's, ...  Returns ``(csr, [canonical...])`` or
        ``None`` if the relations do not share one source type.
        """
        cache = self.__dict__.setdefault("_fused", {})
        if dst_type in cache:
            return cache[dst_type]
        cans = [c for c in self._rels if c[2] == dst_type]
        out = None
        parent = self.__dict__.get("_dropout_parent")
        if parent is not None and all(isinstance(self._rels[c], DroppedRelation) for c in cans):
            base = parent.fused_relations(dst_type)  # built (and validated) once on the parent
            if base is not None and base[1] == cans:
                rels = [self._rels[c] for c in cans]
                if all(r.desc is not None for r in rels) and len(rels) <= 8:
                    # relation i's edges sit at [off_i, off_i + E_i) of the fused edge order: shift the
                    # e_begin / e_end words of its description (offset vectors are cached on the parent)
                    offs = parent.__dict__.setdefault("_fused_offsets", {}).get(dst_type)
                    if offs is None:
                        starts, acc = [], 0
                        for c in cans:
                            starts.append(acc)
                            acc += parent._rels[c].number_of_edges()
                        offs = torch.zeros((len(cans), 8), dtype=torch.int32)
                        offs[:, 0] = offs[:, 1] = torch.tensor(starts, dtype=torch.int32)
                        offs = parent.__dict__["_fused_offsets"][dst_type] = offs.to(rels[0].desc.device)
                    view = base[0].dropped(torch.stack([r.desc for r in rels]) + offs)
                else:
                    view = base[0].masked(torch.cat([r.keep_mask() for r in rels]))
                cache[dst_type] = (view, cans)
                return cache[dst_type]
        if cans and len({c[0] for c in cans}) == 1 and len({self._rels[c].src.device for c in cans}) == 1:
            R = len(cans)
            rels = [self._rels[c] for c in cans]
            n_src, n_dst = rels[0].n_src, rels[0].n_dst
            if R * n_src < 2 ** 31 - 1:
                dst = torch.cat([r.dst.to(torch.int32) for r in rels])
                src = torch.cat([r.src.to(torch.int32) * R + i for i, r in enumerate(rels)])
                csr = ops.CSRGraph(dst, src, n_dst, R * n_src, check_range=not all(r.trusted for r in rels))
                out = (csr, cans)
        cache[dst_type] = out
        return out

    def fused_relations_complement(self, dst_type: str):
        """The fused relations of :meth:`fused_relations` in COMPLEMENT form (f3), or ``None``.

        The reference's encoder graph holds EVERY train pair, both labels (data_loader.py:146-150,170), so
        the label-0 relation of a real dataset is a near-complete bipartite block (lrssl shape: 464 897 of
        519 603 cells) and ``A_0 H = 1 colsum(H)^T - C H`` with ``C`` = the cells NOT in ``A_0`` (the other
        label's pairs and the held-out test pairs: ~55 k) gathers 8x fewer rows.  A block-diagonal union of
        datasets (BASELINE config 3, "Cdataset + Gdataset merged") is near-complete per BLOCK: the identity then
        holds block by block, with one column sum per block (blocks = connected components of relation i0, at most 8).
        Returned ``(csr, cans, i0, blockmat)``: ``csr`` has the fused layout's columns ``R*u + r`` plus one virtual
        source ``R*n_src + b`` per block whose feature row the caller sets to
        ``blockmat[b] @ (scale_i0 * feat_i0)`` (``blockmat``: (B, n_src) 0/1), and signed unit values:

            relation r != i0     its edges,            value +1   (a dropped child: its description)
            complement cells C   (v, R*u + i0),        value -1
            dropped child only   relation i0's edges,  value -1   under its description INVERTED (dgmi_keep.h
                                 kKeepInvert): A_0^kept H = 1 colsum^T - C H - A_0^dropped H — the kernels move
                                 the ~10 % that take part to the front of each id batch, the other 90 % cost
                                 one hash each and no load
            every destination    (v, R*n_src + block(v)),  value +1

        so ``diag(ci) csr diag(scale_ext) feat_ext`` is the hetero-sum aggregate exactly (up to fp32
        summation order).  Taken when relation i0 covers at least ``COMPLEMENT_MIN_DENSITY`` of the cells
        and has no repeated cell (``A_0`` must be 0/1 for the identity to hold; one readback at build)."""
        cache = self.__dict__.setdefault("_fused_complement", {})
        if dst_type in cache:
            return cache[dst_type]
        out = None
        parent = self.__dict__.get("_dropout_parent")
        fr = self.fused_relations(dst_type)
        if fr is not None and parent is not None and all(isinstance(self._rels[c], DroppedRelation) for c in fr[1]):
            struct = parent._complement_struct(dst_type, with_dense_relation=True)
            if struct is not None:
                csr, cans, i0, blockmat, starts = struct
                rels = [self._rels[c] for c in cans]
                if all(r.desc is not None for r in rels) and len(rels) <= 8:
                    key = ("_complement_offsets", dst_type)
                    offs = parent.__dict__.get(key)
                    if offs is None:
                        offs = torch.zeros((len(cans), 8), dtype=torch.int32)
                        offs[:, 0] = offs[:, 1] = torch.tensor(starts, dtype=torch.int32)
                        offs[i0, 6] = 1  # kKeepInvert: relation i0's DROPPED edges are the ones subtracted
                        offs = parent.__dict__[key] = offs.to(rels[0].desc.device)
                    view = csr.dropped(torch.stack([r.desc for r in rels]) + offs)
                else:  # 0/1 masks (the reference's literal randperm selection)
                    parts = [(1.0 - r.keep_mask()) if i == i0 else r.keep_mask() for i, r in enumerate(rels)]
                    tail = torch.ones(csr.nnz - sum(int(p.numel()) for p in parts), dtype=torch.float32, device=csr.device)
                    view = csr.masked(torch.cat(parts + [tail]))
                out = (view, cans, i0, blockmat)
        elif fr is not None and parent is None and not any(isinstance(r, DroppedRelation) for r in self._rels.values()):
            struct = self._complement_struct(dst_type, with_dense_relation=False)
            if struct is not None:
                out = struct[:4]
        cache[dst_type] = out
        return out

    def _complement_struct(self, dst_type: str, with_dense_relation: bool):
        """``(csr, cans, i0, blockmat, starts)`` of the complement form on THIS (un-dropped) graph; ``with_dense_relation``:
        also carry relation i0's own edges (value -1) for dropped children to subtract their dropped part."""
        cache = self.__dict__.setdefault("_complement_structs", {})
        key = (dst_type, with_dense_relation)
        if key in cache:
            return cache[key]
        out = None
        fr = self.fused_relations(dst_type)
        if fr is not None and COMPLEMENT_MIN_DENSITY <= 1.0:
            cans = fr[1]
            rels = [self._rels[c] for c in cans]
            R, n_src, n_dst = len(cans), rels[0].n_src, rels[0].n_dst
            cells = n_src * n_dst
            i0 = max(range(R), key=lambda i: rels[i].number_of_edges())
            E0 = rels[i0].number_of_edges()
            if 0 < cells <= COMPLEMENT_MAX_CELLS and E0 >= COMPLEMENT_MIN_DENSITY * 0.25 * cells and R * n_src + 8 < 2 ** 31 - 1:
                dev = rels[i0].src.device
                present = torch.zeros((n_dst, n_src), dtype=torch.bool, device=dev)
                present[rels[i0].dst.long(), rels[i0].src.long()] = True
                blocks = _bipartite_blocks(present) if int(present.sum()) == E0 else None  # repeated cell: A_i0 is not 0/1
                if blocks is not None:
                    blk_dst, blk_src, B = blocks  # block id per node (-1: no cell of relation i0), B <= 8 blocks
                    support = (blk_dst.view(-1, 1) == blk_src.view(1, -1)) & (blk_dst.view(-1, 1) >= 0)
                    if E0 >= COMPLEMENT_MIN_DENSITY * int(support.sum()):
                        cv, cu = torch.nonzero(support & ~present, as_tuple=True)  # complement cells, row-major
                        dst_parts, col_parts, val_parts, starts, acc = [], [], [], [], 0
                        for i, r in enumerate(rels):
                            starts.append(acc)
                            if i == i0 and not with_dense_relation:
                                continue
                            dst_parts.append(r.dst.to(torch.int32))
                            col_parts.append(r.src.to(torch.int32) * R + i)
                            val_parts.append(torch.full((r.number_of_edges(),), -1.0 if i == i0 else 1.0, device=dev))
                            acc += r.number_of_edges()
                        has = torch.nonzero(blk_dst >= 0, as_tuple=True)[0]  # destinations with a block: one edge to its column sum
                        dst_parts += [cv.to(torch.int32), has.to(torch.int32)]
                        col_parts += [(cu * R + i0).to(torch.int32), (R * n_src + blk_dst[has]).to(torch.int32)]
                        val_parts += [torch.full((cv.numel(),), -1.0, device=dev), torch.ones(has.numel(), device=dev)]
                        csr = ops.CSRGraph(torch.cat(dst_parts), torch.cat(col_parts), n_dst, R * n_src + B,
                                           vals=torch.cat(val_parts), check_range=not all(r.trusted for r in rels))
                        # (B, n_src) 0/1: block b's column sum is blockmat[b] @ (scale_i0 * feat_i0)
                        blockmat = (blk_src.view(1, -1) == torch.arange(B, device=dev).view(-1, 1)).to(torch.float32)
                        out = (csr, cans, i0, blockmat, starts)
        cache[key] = out
        return out

    def edge_pairs(self, etype=None) -> "ops.EdgePairs":
        """Kernel-side view of one relation's edge list for the decoder gather-concat."""
        rel = self[etype] if etype is not None else self._single()
        if getattr(rel, "_pairs", None) is None:
            rel._pairs = ops.EdgePairs(rel.src, rel.dst, rel.n_src, rel.n_dst, check_range=not rel.trusted)
        return rel._pairs

    def _single(self) -> RelationGraph:
        if len(self._rels) != 1:
            raise ValueError("graph has %d edge types; name one" % len(self._rels))
        return next(iter(self._rels.values()))


# ---------------------------------------------------------------------------------------------
# builders for the data formats either side of the path
# ---------------------------------------------------------------------------------------------
def _etype_name(rating) -> str:
    if rating == 0:
        return "0"
    if rating == 1:
        return "1"
    return str(rating).replace(".", "_")


def build_enc_graph(drug_ids: torch.Tensor, dis_ids: torch.Tensor, values: torch.Tensor,
                    n_drug: int, n_dis: int, symm: bool = True, add_support: bool = True,
                    device=None) -> HeteroGraph:
    """Encoder graph in the reference's format — data_loader.py:400-490.

    One relation pair per distinct rating value (``'0'``/``'rev-0'``, ``'1'``/``'rev-1'``),
    ``ci = 1/sqrt(total in-degree)`` and, with ``symm``, ``cj = 1/sqrt(total out-degree)``;
    isolated nodes get 0 (data_loader.py:454-457).
    """
    drug_ids = torch.as_tensor(drug_ids)
    dis_ids = torch.as_tensor(dis_ids)
    values = torch.as_tensor(values)
    if device is not None:
        drug_ids, dis_ids, values = drug_ids.to(device), dis_ids.to(device), values.to(device)
    data = {}
    for rating in torch.unique(values).tolist():
        sel = (values == rating).nonzero(as_tuple=True)[0]
        name = _etype_name(int(rating) if float(rating).is_integer() else rating)
        d, s = drug_ids[sel], dis_ids[sel]
        data[("drug", name, "disease")] = (d, s)
        data[("disease", "rev-" + name, "drug")] = (s, d)
    g = HeteroGraph(data, {"drug": n_drug, "disease": n_dis})
    if add_support:
        def calc_norm(deg):
            deg = deg.to(torch.float32)
            deg = torch.where(deg == 0, torch.full_like(deg, float("inf")), deg)
            return (1.0 / torch.sqrt(deg)).unsqueeze(1)

        drug_deg = torch.bincount(drug_ids.long(), minlength=n_drug)
        dis_deg = torch.bincount(dis_ids.long(), minlength=n_dis)
        drug_ci, dis_ci = calc_norm(drug_deg), calc_norm(dis_deg)
        if symm:
            drug_cj, dis_cj = calc_norm(drug_deg), calc_norm(dis_deg)
        else:
            drug_cj = torch.ones(n_drug, device=drug_ids.device)
            dis_cj = torch.ones(n_dis, device=drug_ids.device)
        g.nodes["drug"].data.update({"ci": drug_ci, "cj": drug_cj})
        g.nodes["disease"].data.update({"ci": dis_ci, "cj": dis_cj})
    return g


def build_dec_graph(drug_ids: torch.Tensor, dis_ids: torch.Tensor, n_drug: int, n_dis: int,
                    device=None) -> HeteroGraph:
    """Decoder graph — data_loader.py:492-509: one relation ``('drug','rate','disease')``."""
    return HeteroGraph({("drug", "rate", "disease"): (torch.as_tensor(drug_ids), torch.as_tensor(dis_ids))},
                       {"drug": n_drug, "disease": n_dis}, device=device)


def from_dgl(g) -> HeteroGraph:
    """Convert a DGL heterograph (or anything with its accessor surface: ``canonical_etypes``,
    ``ntypes``, ``number_of_nodes(nt)``, ``edges(etype=...)``, ``nodes[nt].data``) into a
    :class:`HeteroGraph`, keeping edge order and node data (``ci``/``cj``).

    DGL is not installable in this pipeline, so this is exercised against the accessor-compatible
    stand-in used for fixture generation only (tests/test_host_logic.py); with a real DGL graph it
    relies on those five public accessors and nothing else.
    """
    data = {}
    for can in g.canonical_etypes:
        src, dst = g.edges(etype=can)
        data[tuple(can)] = (src, dst)
    out = HeteroGraph(data, {nt: int(g.number_of_nodes(nt)) for nt in g.ntypes})
    for nt in g.ntypes:
        for key, val in g.nodes[nt].data.items():
            out.nodes[nt].data[key] = val
    return out


def _bipartite_blocks(present: torch.Tensor, max_blocks: int = 8):
    """Connected components of the bipartite graph whose cells are ``present`` (n_dst, n_src bool):
    ``(block_of_dst, block_of_src, B)`` with ids 0..B-1 (-1 for nodes without a cell), or ``None`` when there are
    more than ``max_blocks`` components.  Min-label propagation over the dense map: a near-complete block
    converges in two sweeps; a few host readbacks, at graph build only."""
    n_dst, n_src = present.shape
    big = n_src + n_dst
    lab_src = torch.arange(n_src, device=present.device, dtype=torch.int32)
    lab_dst = torch.full((n_dst,), big, device=present.device, dtype=torch.int32)
    fill = torch.full((1, 1), big, dtype=torch.int32, device=present.device)
    for _ in range(64):
        new_dst = torch.where(present, lab_src.view(1, -1), fill).min(dim=1).values
        new_src = torch.minimum(lab_src, torch.where(present, new_dst.view(-1, 1), fill).min(dim=0).values)
        done = bool(torch.equal(new_src, lab_src) and torch.equal(new_dst, lab_dst))
        lab_src, lab_dst = new_src, new_dst
        if done:
            break
    has_src = present.any(dim=0)
    roots = torch.unique(lab_src[has_src])
    if roots.numel() == 0 or roots.numel() > max_blocks:
        return None
    blk_src = torch.where(has_src, torch.searchsorted(roots, lab_src), torch.full_like(lab_src, -1))
    has_dst = lab_dst < big
    blk_dst = torch.where(has_dst, torch.searchsorted(roots, lab_dst.clamp(max=int(roots.max()))), torch.full_like(lab_dst, -1))
    return blk_dst.to(torch.int64), blk_src.to(torch.int64), int(roots.numel())


def _randperm(n: int, device, generator: Optional[torch.Generator]) -> torch.Tensor:
    """``torch.randperm(n, device=device)`` as the reference draws it (augmentation.py:51,117).  With a CPU generator
    and a GPU graph the permutation is drawn on the HOST and copied over — the generator decides where the draw
    happens, so two runs fed the same CPU generator state see the same subsets whatever device their graphs live on
    (the training-curve parity check of SURVEY 8(d)); no generator / a device generator: the device RNG, as upstream."""
    device = torch.device(device)
    if generator is not None and generator.device.type == "cpu" and device.type != "cpu":
        return torch.randperm(n, generator=generator).to(device)
    return torch.randperm(n, device=device, generator=generator)


def _draw_seed(generator: Optional[torch.Generator]) -> Optional[int]:
    """A 62-bit seed from a CPU generator (or torch's default CPU RNG) without touching the GPU;
    None when the caller handed in a device generator (the randperm path is used then)."""
    if generator is not None and generator.device.type != "cpu":
        return None
    return int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())


def _draw_seeds_on_device(n: int, device, generator: Optional[torch.Generator]) -> torch.Tensor:
    """``n`` 62-bit seeds drawn ON the device (torch's device RNG, or the device generator handed in): nothing
    synchronises and no host value is involved — the form a captured HIP graph needs (``selection="select_device"``).
    The subsets are as uniform as with a host-drawn seed; the RNG stream consumed is the device's, not the CPU's."""
    gen = generator if generator is not None and generator.device.type != "cpu" else None
    return torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device=device, generator=gen)


def _select_kept(E: int, keep: int, device, generator, selection: Optional[str]):
    """(keep_idx or None, desc or None) for a uniformly random subset of exactly ``keep`` edges.

    ``"select"`` (default on the GPU): ``dgmi_random_subset_select`` — radix select on per-edge
    hash keys; the result is the subset's 8-word description, no permutation and no mask is
    materialised.  It consumes ONE ``randint`` draw of the torch CPU generator (the seed), where the
    reference consumes a ``randperm(E)`` — the torch RNG stream after augmentation therefore differs
    from the reference's (a documented deviation; the subset is uniformly random either way).
    ``"randperm"``: the reference's literal procedure, ``torch.randperm(E)[:keep]``
    (augmentation.py:51-52), same RNG consumption as the reference."""
    if selection is None:
        selection = "select" if torch.device(device).type == "cuda" else "randperm"
    if selection == "select_device":
        return None, ops.random_subset_select_batch([E], [keep], _draw_seeds_on_device(1, device, generator), device)[0]
    if selection == "select":
        seed = _draw_seed(generator)
        if seed is not None:
            return None, ops.random_subset_select(E, keep, seed, device)
    return _randperm(E, device, generator)[:keep], None


def random_edge_dropout(graph: HeteroGraph, dropout_rate: float = 0.1,
                        generator: Optional[torch.Generator] = None,
                        selection: Optional[str] = None) -> HeteroGraph:
    """Edge dropout on the encoder graph — augmentation.py:13-89.

    Per edge type independently, keep a uniformly random subset of ``max(1, int(E*(1-p)))`` edges
    (so ``rev-r`` stops being the transpose of ``r``); node data is *copied, not recomputed*
    (augmentation.py:68-70), so ``ci``/``cj`` go stale exactly as in the reference.
    The result holds, per relation, the parent relation and the surviving subset
    (:class:`DroppedRelation`): its products run on the parent's CSRs with a keep mask as edge
    values — nothing is re-sorted and nothing synchronises with the host.
    """
    out = HeteroGraph({}, {nt: graph.number_of_nodes(nt) for nt in graph.ntypes})
    nested = False
    pending = []  # relations whose subset is an on-the-fly description: selected together below
    for can in graph.canonical_etypes:
        rel = graph[can]
        st, _, dt = can
        E = rel.number_of_edges()
        if isinstance(rel, DroppedRelation):  # dropout of a dropout: go through the materialised lists
            nested = True
        if E == 0 or isinstance(rel, DroppedRelation):
            src, dst = rel.src, rel.dst
            if E:
                keep = max(1, int(E * (1 - dropout_rate)))
                perm = _randperm(E, rel.device, generator)[:keep]
                src, dst = src[perm], dst[perm]
            child = RelationGraph(can, src, dst, rel.n_src, rel.n_dst, out._ndata[st], out._ndata[dt])
            child.trusted = True  # a subset of an existing graph's edges: no range re-check, no host sync
            out._rels[can] = child
            continue
        keep = max(1, int(E * (1 - dropout_rate)))
        sel = selection or ("select" if rel.device.type == "cuda" else "randperm")
        if sel == "select_device":  # seeds drawn on the device, all at once, below
            out._rels[can] = None
            pending.append((can, rel, keep, None))
            continue
        seed = _draw_seed(generator) if sel == "select" else None  # one draw per edge type, in canonical order
        if seed is not None:
            out._rels[can] = None  # keeps the canonical position
            pending.append((can, rel, keep, seed))
        else:
            keep_idx = _randperm(E, rel.device, generator)[:keep]
            out._rels[can] = DroppedRelation(rel, keep, out._ndata[st], out._ndata[dt], keep_idx=keep_idx)
    if pending:
        dev0 = pending[0][1].device
        seeds = (_draw_seeds_on_device(len(pending), dev0, generator) if pending[0][3] is None
                 else [s for _, _, _, s in pending])
        descs = ops.random_subset_select_batch([r.number_of_edges() for _, r, _, _ in pending], [k for _, _, k, _ in pending],
                                               seeds, dev0)
        for i, (can, rel, keep, _) in enumerate(pending):
            out._rels[can] = DroppedRelation(rel, keep, out._ndata[can[0]], out._ndata[can[2]], desc=descs[i])
    if not nested:
        out.__dict__["_dropout_parent"] = graph
    for nt in graph.ntypes:
        for k, v in graph.nodes[nt].data.items():
            out.nodes[nt].data[k] = v.clone()
    return out


def random_edge_dropout_sparse(adj, dropout_rate: float = 0.1, generator: Optional[torch.Generator] = None,
                               as_view: bool = False, selection: Optional[str] = None):
    """Edge dropout on a sparse COO adjacency — augmentation.py:92-124.

    Default: a new (uncoalesced, randomly ordered) sparse COO tensor, as the reference returns;
    ``layers.adjacency_csr`` recognises it and applies the dropout as a keep mask over the parent's
    CSR instead of re-sorting its entries.  ``as_view=True`` skips the tensor altogether and returns
    that masked ``CSRGraph`` view directly (``GraphConvolution.forward`` accepts it): no entry is
    copied and nothing synchronises.
    """
    if as_view:
        from .layers import adjacency_csr  # local import: layers imports this module

        base = adjacency_csr(adj)
        alive = base.survivors()
        if alive is not None:
            # dropout of an already dropped view: the reference builds a new tensor from the survivors and
            # keeps max(1, int(E' * (1 - p))) of THOSE (augmentation.py:113-124) — an exact count of E', not
            # an independent subset of the parent's entries.  Rare (one readback), so go through the lists.
            idx = torch.nonzero(alive, as_tuple=False).reshape(-1)
            E = int(idx.numel())
            keep = max(1, int(E * (1 - dropout_rate))) if E else 0
            perm = _randperm(E, base.device, generator)[:keep]
            mask = torch.zeros(base.nnz, dtype=torch.float32, device=base.device).index_fill_(0, idx[perm], 1.0)
            return base.undropped().masked(mask)
        E = base.nnz
        keep = max(1, int(E * (1 - dropout_rate)))
        keep_idx, desc = _select_kept(E, keep, base.device, generator, selection)
        if desc is not None:
            return base.dropped(desc)
        return base.masked(torch.zeros(E, dtype=torch.float32, device=base.device).index_fill_(0, keep_idx, 1.0))
    idx, val = adj._indices(), adj._values()
    E = val.shape[0]
    keep = max(1, int(E * (1 - dropout_rate)))
    perm = _randperm(E, adj.device, generator)[:keep]
    out = torch.sparse_coo_tensor(idx[:, perm], val[perm], adj.shape, device=adj.device)
    out._dgmi_trusted = True  # a subset of a valid adjacency: adjacency_csr skips the id re-check (no host sync)
    # layers.adjacency_csr applies the dropout as a keep mask over the parent's CSR instead of
    # re-sorting these entries (same multiset of entries, hence the same product)
    root = getattr(adj, "_dgmi_parent", None)
    if root is not None:  # a dropout of a dropped tensor: positions are kept relative to the ROOT's entries
        out._dgmi_parent = root
        out._dgmi_keep_idx = adj._dgmi_keep_idx[perm]
    else:
        out._dgmi_parent = adj
        out._dgmi_keep_idx = perm
    return out


def random_edge_dropout_sparse_views(adjs, dropout_rate: float = 0.1, generator: Optional[torch.Generator] = None,
                                     selection: Optional[str] = None):
    """``random_edge_dropout_sparse(..., as_view=True)`` for several adjacencies at once (the four
    similarity / feature graphs of a training step, augmentation.py:448-458): one seed draw per graph,
    in the given order, then ONE batched subset selection.  Returns masked ``CSRGraph`` views."""
    from .layers import adjacency_csr  # local import: layers imports this module

    bases = [adjacency_csr(a) for a in adjs]
    if not bases:
        return []
    on_device = selection == "select_device" and bases[0].device.type == "cuda"
    if selection == "randperm" or (not on_device and (bases[0].device.type != "cuda" or
                                                      (generator is not None and generator.device.type != "cpu"))):
        return [random_edge_dropout_sparse(a, dropout_rate, generator, as_view=True, selection=selection) for a in adjs]
    if any(b.survivors() is not None for b in bases):  # a dropout of dropped views: exact counts of the survivors
        return [random_edge_dropout_sparse(b, dropout_rate, generator, as_view=True) for b in bases]
    keeps = [max(1, int(b.nnz * (1 - dropout_rate))) for b in bases]
    seeds = _draw_seeds_on_device(len(bases), bases[0].device, generator) if on_device else [_draw_seed(generator) for _ in bases]
    descs = ops.random_subset_select_batch([b.nnz for b in bases], keeps, seeds, bases[0].device)
    return [b.dropped(descs[i]) for i, b in enumerate(bases)]


# ---------------------------------------------------------------------------------------------
# (f4) similarity / feature kNN graph construction on the device
# ---------------------------------------------------------------------------------------------
def _coalesce_unit_entries(r: torch.Tensor, c: torch.Tensor, n: int):
    """``sparse_coo_tensor(stack([r, c]), ones).coalesce()`` on the device without the library sort: the entries sorted by
    (row, column) through two passes of the library's own stable record sort (by column, then by row: `dgmi_csr_from_coo_i32`),
    equal neighbours merged, their multiplicity as the float64 value — the same indices and values, bit for bit (4.75 -> ~1.5 ms
    on the 12.9 M entries of a kNN-64 graph of 100 000 nodes)."""
    r32, c32 = r.to(torch.int32), c.to(torch.int32)
    _, r_by_c, e1 = ops.csr_from_coo(c32, r32, n)            # entries in column order: their rows, and where they came from
    c_by_c = c32[e1.long()]
    _, c_sorted, e2 = ops.csr_from_coo(r_by_c, c_by_c, n)    # stable by row: columns ascending inside a row
    r_sorted = r_by_c[e2.long()]
    first = torch.ones(r_sorted.numel(), dtype=torch.bool, device=r.device)
    first[1:] = (r_sorted[1:] != r_sorted[:-1]) | (c_sorted[1:] != c_sorted[:-1])
    st = first.nonzero().squeeze(1)
    counts = torch.diff(st, append=torch.tensor([r_sorted.numel()], device=r.device))
    return torch.stack([r_sorted[st].long(), c_sorted[st].long()]), counts.to(torch.float64)


def _normalized_adjacency(rows: torch.Tensor, cols: torch.Tensor, n: int, symm: bool) -> torch.Tensor:
    """data_loader.py:297-308 + utils.py:11-27 after the neighbour choice: ones COO ->
    (A + A^T if symm) -> + I -> D^-1 (.) -> sparse COO fp32, entries row-major sorted."""
    eye = torch.arange(n, device=rows.device)
    if symm:
        r = torch.cat([rows, cols, eye])
        c = torch.cat([cols, rows, eye])
    else:
        r = torch.cat([rows, eye])
        c = torch.cat([cols, eye])
    if r.is_cuda and n < 2 ** 31 and 2 ** 18 <= r.numel() < 2 ** 31:  # (below, the library's coalesce is as fast: 0.2-0.3 ms)
        idx, val = _coalesce_unit_entries(r, c, n)
    else:
        v = torch.ones(r.numel(), dtype=torch.float64, device=rows.device)
        adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).coalesce()
        idx, val = adj.indices(), adj.values()
    # row sums of the coalesced (row-major sorted) entries as differences of a running sum between the row boundaries: the
    # values are small integers (1 per edge, summed by the coalesce), so every partial sum is exact in float64 and the result
    # is bit for bit that of `index_add_` — whose float64 atomics took 39 ms on the 12.9 M entries of a kNN-64 graph of
    # 100 000 nodes (this form: 0.2 ms)
    bounds = torch.searchsorted(idx[0].contiguous(), torch.arange(n + 1, device=rows.device))
    run = torch.cat([torch.zeros(1, dtype=torch.float64, device=rows.device), val.cumsum(0)])
    rowsum = run[bounds[1:]] - run[bounds[:-1]]
    inv = torch.where(rowsum != 0, 1.0 / rowsum, torch.zeros_like(rowsum))  # utils.py:14-15
    out = torch.sparse_coo_tensor(idx, (val * inv[idx[0]]).to(torch.float32), (n, n))
    out._dgmi_trusted = True
    return out


def similarity_graph(sim: torch.Tensor, k: int, symm: bool = True) -> torch.Tensor:
    """``DrugDataLoader._create_similarity_graph`` (data_loader.py:278-310) on the device.

    The k largest similarities per row (``argpartition(-sim, kth=k)[:, :k]``, usually including
    the node itself) become unit edges; then symmetrise, add I, row-normalise.  Ties at the k-th
    value are broken arbitrarily upstream (argpartition) and here (topk).
    """
    n = sim.shape[0]
    k_actual = min(k, n - 1)
    nbr = torch.topk(sim, k_actual, dim=1).indices
    rows = torch.arange(n, device=sim.device).repeat_interleave(k_actual)
    return _normalized_adjacency(rows, nbr.reshape(-1), n, symm)


def feature_similarity_graph(features: torch.Tensor, k: int, symm: bool = True,
                             block_rows: int = 8192, fused=None) -> torch.Tensor:
    """``_create_feature_similarity_graph`` (data_loader.py:312-344): cosine-similarity kNN graph
    from embeddings; the N x N similarity matrix is never materialised.  On the GPU the neighbour
    search is ``ops.knn_cosine_topk`` when the shape fits it (D % 8 == 0, k <= 16, or k <= 64 from 1536 rows — the reference runs
    k = 4 on 768-d embeddings, train.py:423): fp32-MFMA tiles + on-chip top-k at the reference's sizes,
    a bf16-MFMA screen with exact fp32 rescoring from 1536 rows (1.5x / 13x the torch path at
    N = 763 / 100 000); otherwise, or with ``fused=False``, a row-blocked torch GEMM +
    top-k (an 8192-row block against 100k columns is 3.3 GB of fp32, reduced to k ids per row at
    once).  ``fused=True`` insists on the kernel."""
    n = features.shape[0]
    k_actual = min(k, n - 1)
    norms = features.norm(dim=1, keepdim=True)
    norms = torch.where(norms == 0, torch.full_like(norms, 1e-10), norms)  # data_loader.py:334-335
    xn = features / norms
    if (fused is not False and xn.is_cuda and xn.dtype == torch.float32 and k_actual >= 1
            and ops.knn_cosine_supported(n, xn.shape[1], k_actual)):
        # fused HIP kernel: fp32-MFMA similarity tiles + running top-k on chip (csrc/dgmi_knn.hip)
        nbr = ops.knn_cosine_topk(xn.contiguous(), k_actual).long()
    else:  # widths / k the kernel does not take: blocked GEMM + top-k through torch
        if fused is True:
            raise RuntimeError("fused kNN kernel does not support N=%d D=%d k=%d" % (n, xn.shape[1], k_actual))
        nbr = torch.empty((n, k_actual), dtype=torch.int64, device=features.device)
        for lo in range(0, n, block_rows):
            hi = min(lo + block_rows, n)
            nbr[lo:hi] = torch.topk(xn[lo:hi] @ xn.t(), k_actual, dim=1).indices
    rows = torch.arange(n, device=features.device).repeat_interleave(k_actual)
    return _normalized_adjacency(rows, nbr.reshape(-1), n, symm)
