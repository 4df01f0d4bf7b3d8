"""CPU oracle for the DREAM-GNN message-passing hot path (TEST INFRASTRUCTURE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
Nothing under ``dream_gnn_amd/`` imports it and there is no CPU fallback in the
product path.

Two layers:

* ctypes wrappers over ``libdgmi_oracle.so`` (``oracle/dgmi_oracle.c``): the
  primitives (stable COO->CSR, CSR SpMM with diagonal scalings, COO SpMM).
* numpy restatements of the reference modules *around* the primitive, each citing
  the reference lines it follows (``/root/reference/layers.py`` etc.).

Parity status: ``th.spmm`` and the reference's own ``utils.py`` functions are pinned
by ``tests/golden``; DGL's ``copy_u -> sum`` is **parity unpinned** (see the header of
``dgmi_oracle.c``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdgmi_oracle.so")
_lib = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile ``libdgmi_oracle.so`` with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "dgmi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "all"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_csr_from_coo_i32.restype = ctypes.c_int
        L.oracle_csr_from_coo_i32.argtypes = [_i32p, _i32p, ctypes.c_int64, ctypes.c_int64, _i32p, _i32p, _i32p]
        for name, yp in (("oracle_spmm_csr_f32", _f32p), ("oracle_spmm_csr_f64", _f64p)):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [_i32p, _i32p, _f32p, _f32p, ctypes.c_int64, _f32p, _f32p, yp,
                           ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.oracle_spmm_csr_abs_f64.restype = None
        L.oracle_spmm_csr_abs_f64.argtypes = [_i32p, _i32p, _f32p, _f32p, ctypes.c_int64, _f32p, _f32p, _f64p,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.oracle_spmm_coo_f32.restype = None
        L.oracle_spmm_coo_f32.argtypes = [_i64p, _i64p, _f32p, ctypes.c_int64, _f32p, ctypes.c_int64, _f32p,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.oracle_gather_f32.restype = None
        L.oracle_gather_f32.argtypes = [_f32p, _i32p, ctypes.c_int64, _f32p]
        L.oracle_gather_concat_f32.restype = None
        L.oracle_gather_concat_f32.argtypes = [_i32p, _i32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int64,
                                               _f32p, ctypes.c_int64, ctypes.c_int64, _f32p, ctypes.c_int64]
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ct)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def max_threads() -> int:
    return int(lib().oracle_max_threads())


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def csr_from_coo(row, col, n_rows):
    """Stable COO->CSR: (indptr, indices, eid).  DGL COOToCSR semantics behind
    ``dgl.heterograph`` (data_loader.py:448, augmentation.py:65)."""
    row = _c(row, np.int32)
    col = _c(col, np.int32)
    E = int(row.shape[0])
    indptr = np.empty(n_rows + 1, np.int32)
    indices = np.empty(E, np.int32)
    eid = np.empty(E, np.int32)
    rc = lib().oracle_csr_from_coo_i32(_p(row, _i32p), _p(col, _i32p), E, int(n_rows),
                                       _p(indptr, _i32p), _p(indices, _i32p), _p(eid, _i32p))
    if rc != 0:
        raise ValueError("row id out of range in csr_from_coo (rc=%d)" % rc)
    return indptr, indices, eid


def csr_sliced_from_coo(row, col, n_rows, n_cols, n_slices):
    """Source-sliced CSR (the XCD-local kernel's layout): stable sort by
    ``slice(col) * n_rows + row`` with ``slice = col // ceil(n_cols / n_slices)``.
    Returns (segptr[n_slices*n_rows+1], indices, eid)."""
    row = np.asarray(row, np.int64)
    col = np.asarray(col, np.int64)
    width = max(1, -(-int(n_cols) // int(n_slices)))
    key = np.minimum(col // width, n_slices - 1) * n_rows + row
    order = np.argsort(key, kind="stable")
    segptr = np.zeros(n_slices * n_rows + 1, np.int64)
    np.add.at(segptr, key + 1, 1)
    return np.cumsum(segptr).astype(np.int32), col[order].astype(np.int32), order.astype(np.int32)


def spmm_csr(indptr, indices, vals, X, src_scale=None, dst_scale=None, threads=1, acc="f32", validate=True):
    """``Y = diag(dst_scale) A diag(src_scale) X`` over a CSR; ``vals=None`` is
    ``update_all(copy_u, sum)`` (layers.py:229-232), else ``th.spmm`` (layers.py:312).
    ``acc``: 'f32' (sequential fp32, CSR order), 'f64' (double), 'abs' (sum |terms|)."""
    indptr = _c(indptr, np.int32)
    indices = _c(indices, np.int32)
    vals = _c(vals, np.float32)
    X = _c(X, np.float32)
    src_scale = _c(None if src_scale is None else np.asarray(src_scale).reshape(-1), np.float32)
    dst_scale = _c(None if dst_scale is None else np.asarray(dst_scale).reshape(-1), np.float32)
    n_dst = indptr.shape[0] - 1
    F = X.shape[1]
    if validate and indices.size and (indices.min() < 0 or indices.max() >= X.shape[0]):
        raise ValueError("column id out of range")
    L = lib()
    if acc == "f32":
        Y = np.empty((n_dst, F), np.float32)
        L.oracle_spmm_csr_f32(_p(indptr, _i32p), _p(indices, _i32p), _p(vals, _f32p), _p(X, _f32p), F,
                              _p(src_scale, _f32p), _p(dst_scale, _f32p), _p(Y, _f32p), F, n_dst, F, int(threads))
    elif acc == "f64":
        Y = np.empty((n_dst, F), np.float64)
        L.oracle_spmm_csr_f64(_p(indptr, _i32p), _p(indices, _i32p), _p(vals, _f32p), _p(X, _f32p), F,
                              _p(src_scale, _f32p), _p(dst_scale, _f32p), _p(Y, _f64p), F, n_dst, F, int(threads))
    elif acc == "abs":
        Y = np.empty((n_dst, F), np.float64)
        L.oracle_spmm_csr_abs_f64(_p(indptr, _i32p), _p(indices, _i32p), _p(vals, _f32p), _p(X, _f32p), F,
                                  _p(src_scale, _f32p), _p(dst_scale, _f32p), _p(Y, _f64p), F, n_dst, F)
    else:
        raise ValueError(acc)
    return Y


def spmm_coo(row, col, vals, X, n_rows):
    """``th.spmm`` on an uncoalesced COO, entries applied in storage order (layers.py:312)."""
    row = _c(row, np.int64)
    col = _c(col, np.int64)
    vals = _c(vals, np.float32)
    X = _c(X, np.float32)
    F = X.shape[1]
    Y = np.empty((n_rows, F), np.float32)
    lib().oracle_spmm_coo_f32(_p(row, _i64p), _p(col, _i64p), _p(vals, _f32p), row.shape[0], _p(X, _f32p), F,
                              _p(Y, _f32p), F, int(n_rows), F)
    return Y


def gather_concat(src, dst, A, B):
    """``cat(A[src[e]], B[dst[e]])`` per edge — udf_u_mul_e (layers.py:364,378-379)."""
    src = _c(src, np.int32)
    dst = _c(dst, np.int32)
    A = _c(A, np.float32)
    B = _c(B, np.float32)
    E, Fa, Fb = src.shape[0], A.shape[1], B.shape[1]
    out = np.empty((E, Fa + Fb), np.float32)
    lib().oracle_gather_concat_f32(_p(src, _i32p), _p(dst, _i32p), E, _p(A, _f32p), Fa, Fa, _p(B, _f32p), Fb, Fb,
                                   _p(out, _f32p), Fa + Fb)
    return out


def transpose_coo(row, col):
    """Edges of the reversed graph (backward of copy_u->sum is copy_u->sum on it)."""
    return np.asarray(col), np.asarray(row)


# --------------------------------------------------------------------------
# graph-side restatements (data formats either side of the path)
# --------------------------------------------------------------------------
def enc_graph_norms(drug_ids, dis_ids, values, n_drug, n_dis, symm=True):
    """``ci``/``cj`` node data of the encoder graph — data_loader.py:453-488.

    ci[node] = 1/sqrt(sum_r in_degree_r(node)), cj likewise from out-degrees when
    ``symm``; a node with no edge gets 0 (x==0 -> inf -> 1/sqrt(inf) = 0, :455-457).
    Returns dict of (N,1) float32 arrays: drug_ci, drug_cj, dis_ci, dis_cj.
    """
    drug_ids = np.asarray(drug_ids)
    dis_ids = np.asarray(dis_ids)
    del values  # every rating's edges are summed (:462-478)
    drug_deg = np.bincount(drug_ids, minlength=n_drug).astype(np.float32)
    dis_deg = np.bincount(dis_ids, minlength=n_dis).astype(np.float32)

    def _calc_norm(x):
        x = x.astype(np.float32).copy()
        x[x == 0.0] = np.inf
        return (1.0 / np.sqrt(x)).astype(np.float32).reshape(-1, 1)

    out = {"drug_ci": _calc_norm(drug_deg), "dis_ci": _calc_norm(dis_deg)}
    if symm:
        out["drug_cj"] = _calc_norm(drug_deg)
        out["dis_cj"] = _calc_norm(dis_deg)
    else:  # data_loader.py:483-485 (1-D ones; layers.py:224 views them (-1,1))
        out["drug_cj"] = np.ones((n_drug,), np.float32)
        out["dis_cj"] = np.ones((n_dis,), np.float32)
    return out


def normalize_rows(dense_or_csr):
    """utils.normalize (utils.py:11-17) on a scipy sparse matrix."""
    import scipy.sparse as sp

    mx = sp.csr_matrix(dense_or_csr)
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def similarity_graph_coo(sim, k, symm=True):
    """_create_similarity_graph (data_loader.py:278-310) -> (row i64, col i64, val f32, N)."""
    import scipy.sparse as sp

    n = sim.shape[0]
    k_actual = min(k, n - 1)
    neighbor = np.argpartition(-sim, kth=k_actual, axis=1)[:, :k_actual]
    row_index = np.arange(n).repeat(neighbor.shape[1])
    col_index = neighbor.reshape(-1)
    adj = sp.coo_matrix((np.ones(len(row_index)), (row_index, col_index)), shape=(n, n))
    if symm:
        adj = adj + adj.T
        adj = adj.multiply(adj > 0)
    norm = normalize_rows(adj + sp.eye(n)).tocoo().astype(np.float32)
    return norm.row.astype(np.int64), norm.col.astype(np.int64), norm.data.astype(np.float32), n


def _edge_hash(seed, E):
    """hash32(seed, e) for e in [0, E): restatement of ``edge_hash`` in dream_gnn_amd/csrc/dgmi_keep.h — two 32-bit
    avalanche rounds (murmur3's finaliser, then lowbias32), keyed with the low / high half of the seed; a bijection of
    the edge id for a fixed seed."""
    seed &= 0xFFFFFFFFFFFFFFFF
    m = np.uint64(0xFFFFFFFF)
    x = (np.arange(E, dtype=np.uint64) ^ np.uint64(seed & 0xFFFFFFFF)) & m  # 32-bit arithmetic carried in uint64
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & m
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & m
    x ^= x >> np.uint64(16)
    x = (x + np.uint64(seed >> 32)) & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x


def random_subset_mask(E, keep, seed):
    """Restatement of ``dgmi_random_subset_mask_f32``: keys (hash32(seed, e), e); the ``keep`` smallest keys are kept.  The subset it stands for is the
    reference's ``randperm(E)[:num_keep]`` (augmentation.py:48-52): uniformly random, exact size."""
    h = _edge_hash(seed, E)
    order = np.lexsort((np.arange(E), h))  # by hash, ties by edge id
    mask = np.zeros(E, np.float32)
    mask[order[:keep]] = 1.0
    return mask


def random_subset_select(E, keep, seed, e_offset=0):
    """Restatement of ``dgmi_random_subset_select``: the 8-word description of the same subset —
    {e_begin, e_end, seed_lo, seed_hi, threshold hash, tie cut, 0, 0} as int32 bit patterns:
    edge ``e_offset + i`` is kept iff ``hash(i) < thr or (hash(i) == thr and i <= tie_cut)``."""
    h = _edge_hash(seed, E)
    thr, tie_cut = 0, -1
    if keep > 0:
        order = np.lexsort((np.arange(E), h))
        last = int(order[keep - 1])
        thr = int(h[last])
        ties_kept = keep - int((h < thr).sum())
        tie_ids = np.flatnonzero(h == thr)
        tie_cut = int(tie_ids[ties_kept - 1]) if ties_kept > 0 else -1
        assert tie_cut == last or ties_kept == 0
    seed &= 0xFFFFFFFFFFFFFFFF
    words = np.array([e_offset, e_offset + E, seed & 0xFFFFFFFF, seed >> 32, thr, tie_cut & 0xFFFFFFFF, 0, 0], np.uint64)
    return words.astype(np.uint32).view(np.int32)


def keep_mask(desc, E):
    """Restatement of ``dgmi_keep_mask_f32``: float 0/1 over edges [0, E) under (n, 8) descriptions;
    edges outside every description are kept, an edge covered by several descriptions survives only
    if every one of them keeps it (csrc/dgmi_keep.h edge_kept)."""
    d = np.asarray(desc, np.int32).reshape(-1, 8).view(np.uint32).astype(np.uint64)
    mask = np.ones(E, np.float32)
    for e_begin, e_end, lo, hi, thr, cut, flags, _ in d:
        e_begin, e_end = int(e_begin), min(int(e_end), E)
        if e_end <= e_begin:
            continue
        n = e_end - e_begin
        h = _edge_hash((int(hi) << 32) | int(lo), n)
        cut = int(np.uint32(cut).view(np.int32)) if isinstance(cut, np.generic) else int(np.array(cut, np.uint32).view(np.int32))
        kept = (h < thr) | ((h == thr) & (np.arange(n) <= cut))
        if int(flags) & 1:  # inverted description (kKeepInvert): the edges it drops take part
            kept = ~kept
        mask[e_begin:e_end] *= kept.astype(np.float32)  # intersection over the covering descriptions
    return mask


def compact_layout(ptr, indices, vals, eid, desc):
    """Restatement of ``dgmi_compact_layout_i32``: a CSR-shaped layout of the parent graph with the edges dropped
    under the description(s) ``desc`` removed, stable — i.e. the same layout of the graph the reference REBUILDS from
    the kept edges each iteration (augmentation.py:48-65 ``dgl.heterograph`` of ``src[keep], dst[keep]``; :114-124
    for a sparse COO).  Returns (ptr_out, indices_out, vals_out): only the survivors, not padded."""
    ptr, indices, eid = np.asarray(ptr, np.int64), np.asarray(indices), np.asarray(eid, np.int64)
    kept = keep_mask(desc, int(eid.max()) + 1 if eid.size else 0)[eid].astype(bool) if eid.size else np.zeros(0, bool)
    rank = np.concatenate([[0], np.cumsum(kept)])
    return rank[ptr].astype(np.int32), indices[kept], None if vals is None else np.asarray(vals)[kept]


def edge_dropout_keep(num_edges, dropout_rate, perm):
    """Kept edge positions of random_edge_dropout(_sparse): the first
    max(1, int(E*(1-p))) entries of a permutation (augmentation.py:48-52,114-118)."""
    num_keep = max(1, int(num_edges * (1 - dropout_rate)))
    return np.asarray(perm)[:num_keep]


# --------------------------------------------------------------------------
# module-level restatements (numpy, eval-mode arithmetic; dropout masks are inputs)
# --------------------------------------------------------------------------
def gcmc_graph_conv(src, dst, n_src, n_dst, feat, weight, cj, ci, cj_mask=None):
    """GCMCGraphConv.forward — layers.py:169-236.

    feat@W (:220-221) -> * dropout(cj) (:224-225) -> copy_u/sum (:229-232) -> * ci (:234).
    ``cj_mask`` is the (N_src,1) multiplicative dropout mask (already scaled by 1/(1-p)),
    None in eval mode.
    """
    h = feat if weight is None else (np.asarray(feat, np.float32) @ np.asarray(weight, np.float32))
    cjd = np.asarray(cj, np.float32).reshape(-1, 1)
    if cj_mask is not None:
        cjd = cjd * np.asarray(cj_mask, np.float32).reshape(-1, 1)
    h = (h * cjd).astype(np.float32)
    indptr, indices, _ = csr_from_coo(dst, src, n_dst)
    rst = spmm_csr(indptr, indices, None, h)
    del n_src
    return (rst * np.asarray(ci, np.float32).reshape(-1, 1)).astype(np.float32)


def _leaky(x, slope=0.1):
    return np.where(x >= 0, x, slope * x).astype(np.float32)


def gcmc_layer(rels, n_drug, n_dis, drug_feat, dis_feat, params, norms, rating_vals=(0, 1),
               share=True, agg_act=None):
    """GCMCLayer.forward (eval mode) — layers.py:117-143.

    ``rels[rating] = (drug_ids, dis_ids)`` edges of relation ``rating`` (drug -> disease);
    the reverse relation is the transposed list (data_loader.py:443-446).
    ``params``: att (R,B), basis (B,in,msg), ufc_w (out,msg), ufc_b, [ifc_w, ifc_b],
    or per-etype own weights ``w_<etype>`` when not sharing (layers.py:86-97).
    Hetero aggregate = sum over relations per destination type (layers.py:98,129).
    """
    att = np.asarray(params["att"], np.float32)
    basis = np.asarray(params["basis"], np.float32)
    B, fin, msg = basis.shape
    W = (att @ basis.reshape(B, -1)).reshape(-1, fin, msg)  # layers.py:120-121
    dis_out = np.zeros((n_dis, msg), np.float32)
    drug_out = np.zeros((n_drug, msg), np.float32)
    for i, r in enumerate(rating_vals):
        name = str(r).replace(".", "_")
        d_ids, s_ids = rels[r]
        w_fwd = W[i] if share else np.asarray(params["w_" + name], np.float32)
        w_rev = W[i] if share else np.asarray(params["w_rev-" + name], np.float32)
        # ('drug', r, 'disease'): messages drug -> disease
        dis_out = dis_out + gcmc_graph_conv(d_ids, s_ids, n_drug, n_dis, drug_feat, w_fwd,
                                            norms["drug_cj"], norms["dis_ci"])
        # ('disease', rev-r, 'drug')
        drug_out = drug_out + gcmc_graph_conv(s_ids, d_ids, n_dis, n_drug, dis_feat, w_rev,
                                              norms["dis_cj"], norms["drug_ci"])
    act = (lambda x: x) if agg_act is None else agg_act
    drug_out, dis_out = act(drug_out), act(dis_out)  # :134-138 (dropout = identity in eval)
    ifc_w = params.get("ifc_w", params["ufc_w"])
    ifc_b = params.get("ifc_b", params["ufc_b"])
    drug_o = drug_out @ np.asarray(ifc_w, np.float32).T + np.asarray(ifc_b, np.float32)  # :140
    dis_o = dis_out @ np.asarray(params["ufc_w"], np.float32).T + np.asarray(params["ufc_b"], np.float32)  # :141
    return drug_o.astype(np.float32), dis_o.astype(np.float32)


def graph_convolution(x, adj_row, adj_col, adj_val, n, weight, bias=None):
    """GraphConvolution.forward — layers.py:306-316: spmm(adj, x@W) + b."""
    support = np.asarray(x, np.float32) @ np.asarray(weight, np.float32)
    out = spmm_coo(adj_row, adj_col, adj_val, support, n)
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return out.astype(np.float32)


def gcn(x, adj, params):
    """GCN.forward (eval) — layers.py:245-249: gc2(relu(gc1(x, adj)), adj)."""
    r, c, v, n = adj
    h = np.maximum(graph_convolution(x, r, c, v, n, params["gc1.weight"], params["gc1.bias"]), 0)
    return graph_convolution(h, r, c, v, n, params["gc2.weight"], params["gc2.bias"])
