"""In-memory stand-in for the ``dgl`` import of the reference — FIXTURE GENERATION ONLY.

DGL is not installed in the build container and cannot be (no network, no wheel), and the
reference does not pin or vendor it.  ``gen_golden.py`` registers this module as ``dgl`` in
``sys.modules`` so that the reference's *own* ``layers.py`` / ``augmentation.py`` /
``data_loader.py`` import and run unmodified from ``/root/reference``; the golden vectors then
pin everything the reference's code does AROUND the DGL primitive (weight composition,
scaling order, dropout placement, hetero sum, output routing, edge-dropout bookkeeping).

What this does NOT do: it is not DGL.  ``update_all(copy_u, sum)`` below is our statement of
DGL's documented semantics (sum over in-edges, multigraph, zero rows for isolated nodes),
written with ``torch.index_add_`` so that it shares no code with ``oracle/``.  DGL's own kernel
therefore stays **parity unpinned** — see DESIGN.md.

Never imported by the product or on the GPU box; besides gen_golden.py only one CPU
host-logic test uses its graph class, as a DGL-*shaped* object to feed `graph.from_dgl`.
"""
from __future__ import annotations

import contextlib
import sys
import types

import torch


class DGLError(Exception):
    pass


# ---- dgl.function ---------------------------------------------------------------------------
class _Msg:
    def __init__(self, kind, src_field, out_field):
        self.kind, self.src_field, self.out_field = kind, src_field, out_field


class _Red:
    def __init__(self, kind, msg_field, out_field):
        self.kind, self.msg_field, self.out_field = kind, msg_field, out_field


def copy_u(u, out):
    return _Msg("copy_u", u, out)


def fsum(msg, out):
    return _Red("sum", msg, out)


# ---- graph objects --------------------------------------------------------------------------
class _View:
    def __init__(self, data):
        self.data = data


class _Indexer:
    def __init__(self, fn):
        self._fn = fn

    def __getitem__(self, key):
        return self._fn(key)


class _Edges:
    pass


class DGLGraph:
    """Heterograph (possibly with one relation = a relation slice)."""

    def __init__(self, data_dict, num_nodes_dict, ndata=None, idtype=torch.int64):
        # DGL's ``heterograph`` sorts node types and relation triples ("to have a deterministic
        # order for the same set of type names", dgl/convert.py — from memory of DGL's public
        # source, unverifiable here): canonical_etypes / ntypes iterate in sorted order, which is
        # the order HeteroGraphConv runs the sub-modules (their dropout draws) and the order
        # augmentation.py:24 walks the edge types.
        self._nn = {nt: num_nodes_dict[nt] for nt in sorted(num_nodes_dict)}
        self._ndata = ndata if ndata is not None else {nt: {} for nt in self._nn}
        self._rel = {}
        self._edata = {}
        for can in sorted(data_dict):
            s, d = data_dict[can]
            s = torch.as_tensor(s).to(idtype)
            d = torch.as_tensor(d).to(idtype)
            self._rel[can] = (s, d)
            self._edata[can] = {}
        self.idtype = idtype

    # structure
    @property
    def ntypes(self):
        return list(self._nn)

    @property
    def canonical_etypes(self):
        return list(self._rel)

    @property
    def etypes(self):
        return [c[1] for c in self._rel]

    @property
    def device(self):
        for s, _ in self._rel.values():
            return s.device
        return torch.device("cpu")

    def _can(self, key):
        if key is None:
            assert len(self._rel) == 1
            return next(iter(self._rel))
        if isinstance(key, tuple):
            return key
        hits = [c for c in self._rel if c[1] == key]
        assert len(hits) == 1, key
        return hits[0]

    def number_of_nodes(self, ntype=None):
        return self._nn[ntype]

    def number_of_edges(self, etype=None):
        if etype is None and len(self._rel) != 1:
            return sum(s.shape[0] for s, _ in self._rel.values())
        return self._rel[self._can(etype)][0].shape[0]

    @property
    def edges(self):
        """``g.edges(etype=...)`` and ``g.edges[etype].data`` share one name in DGL."""
        return _EdgesAccessor(self)

    def __getitem__(self, key):
        can = self._can(key)
        st, _, dt = can
        g = DGLGraph({can: self._rel[can]}, {st: self._nn[st], dt: self._nn[dt]},
                     ndata={st: self._ndata[st], dt: self._ndata[dt]}, idtype=self.idtype)
        return g

    def int(self):
        return DGLGraph({c: (s.int(), d.int()) for c, (s, d) in self._rel.items()}, self._nn,
                        ndata=self._ndata, idtype=torch.int32)

    def to(self, device):
        g = DGLGraph({c: (s.to(device), d.to(device)) for c, (s, d) in self._rel.items()}, self._nn,
                     ndata={nt: {k: v.to(device) for k, v in d.items()} for nt, d in self._ndata.items()},
                     idtype=self.idtype)
        return g

    def clone(self):
        return DGLGraph({c: (s.clone(), d.clone()) for c, (s, d) in self._rel.items()}, self._nn,
                        ndata={nt: {k: v.clone() for k, v in d.items()} for nt, d in self._ndata.items()},
                        idtype=self.idtype)

    # data views
    @property
    def nodes(self):
        return _Indexer(lambda nt: _View(self._ndata[nt]))

    @property
    def srcdata(self):
        return self._ndata[self._can(None)[0]]

    @property
    def dstdata(self):
        return self._ndata[self._can(None)[2]]

    @property
    def edata(self):
        return self._edata[self._can(None)]

    def number_of_src_nodes(self):
        return self._nn[self._can(None)[0]]

    def number_of_dst_nodes(self):
        return self._nn[self._can(None)[2]]

    def in_degrees(self):
        _, d = self._rel[self._can(None)]
        return torch.bincount(d.long(), minlength=self.number_of_dst_nodes())

    def out_degrees(self):
        s, _ = self._rel[self._can(None)]
        return torch.bincount(s.long(), minlength=self.number_of_src_nodes())

    @contextlib.contextmanager
    def local_scope(self):
        saved_n = {nt: dict(d) for nt, d in self._ndata.items()}
        saved_e = {c: dict(d) for c, d in self._edata.items()}
        try:
            yield
        finally:
            for nt, d in self._ndata.items():
                d.clear()
                d.update(saved_n[nt])
            for c, d in self._edata.items():
                d.clear()
                d.update(saved_e[c])

    # message passing
    def update_all(self, msg, red):
        assert msg.kind == "copy_u" and red.kind == "sum" and msg.out_field == red.msg_field
        s, d = self._rel[self._can(None)]
        h = self.srcdata[msg.src_field]
        out = torch.zeros((self.number_of_dst_nodes(),) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
        out = out.index_add(0, d.long(), h.index_select(0, s.long()))
        self.dstdata[red.out_field] = out

    def apply_edges(self, udf, etype=None):
        can = self._can(etype)
        s, d = self._rel[can]
        e = _Edges()
        e.src = {k: v.index_select(0, s.long()) for k, v in self._ndata[can[0]].items()}
        e.dst = {k: v.index_select(0, d.long()) for k, v in self._ndata[can[2]].items()}
        self._edata[can].update(udf(e))


class _EdgesAccessor:
    def __init__(self, g):
        self._g = g

    def __call__(self, etype=None):
        return self._g._rel[self._g._can(etype)]

    def __getitem__(self, et):
        return _View(self._g._edata[self._g._can(et)])


def heterograph(data_dict, num_nodes_dict=None, idtype=torch.int64, device=None):
    g = DGLGraph(data_dict, num_nodes_dict, idtype=idtype)
    return g if device is None else g.to(device)


def bipartite_from_scipy(sp_mat, utype, etype, vtype):
    coo = sp_mat.tocoo()
    return DGLGraph({(utype, etype, vtype): (torch.from_numpy(coo.row.astype("int64")),
                                             torch.from_numpy(coo.col.astype("int64")))},
                    {utype: sp_mat.shape[0], vtype: sp_mat.shape[1]})


# ---- dgl.nn.pytorch.HeteroGraphConv ----------------------------------------------------------
class HeteroGraphConv(torch.nn.Module):
    def __init__(self, mods, aggregate="sum"):
        super().__init__()
        self.mods = torch.nn.ModuleDict(mods)
        self.aggregate = aggregate

    def forward(self, g, inputs, mod_args=None, mod_kwargs=None):
        mod_args = mod_args or {}
        mod_kwargs = mod_kwargs or {}
        outputs = {nty: [] for nty in g.ntypes}
        for stype, etype, dtype in g.canonical_etypes:
            rel_graph = g[stype, etype, dtype]
            if stype not in inputs:
                continue
            dstdata = self.mods[etype](rel_graph, (inputs[stype], inputs[dtype]),
                                       *mod_args.get(etype, ()), **mod_kwargs.get(etype, {}))
            outputs[dtype].append(dstdata)
        rsts = {}
        for nty, alist in outputs.items():
            if len(alist) != 0:
                if self.aggregate == "sum":
                    rsts[nty] = torch.stack(alist, 0).sum(0)
                elif self.aggregate == "stack":
                    rsts[nty] = torch.stack(alist, 1)
                else:
                    raise NotImplementedError(self.aggregate)
        return rsts


def install():
    """Register the stand-in as ``dgl`` (+ submodules the reference imports)."""
    dgl = types.ModuleType("dgl")
    dgl.DGLError = DGLError
    dgl.DGLGraph = DGLGraph
    dgl.heterograph = heterograph
    dgl.bipartite_from_scipy = bipartite_from_scipy
    fn = types.ModuleType("dgl.function")
    fn.copy_u = copy_u
    fn.sum = fsum
    dgl.function = fn
    dgl.fn = fn
    nn_mod = types.ModuleType("dgl.nn")
    nn_pt = types.ModuleType("dgl.nn.pytorch")
    nn_pt.HeteroGraphConv = HeteroGraphConv
    nn_mod.pytorch = nn_pt
    dgl.nn = nn_mod
    sys.modules["dgl"] = dgl
    sys.modules["dgl.function"] = fn
    sys.modules["dgl.fn"] = fn
    sys.modules["dgl.nn"] = nn_mod
    sys.modules["dgl.nn.pytorch"] = nn_pt
    return dgl
