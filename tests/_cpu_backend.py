"""Oracle-backed CPU stand-ins for the three ops, for HOST-LOGIC tests only.

``-m "not gpu"`` tests exercise the drop-in modules' Python (weight composition, scaling and
dropout placement, hetero sum, state_dict layout, caching, autograd wiring) on the build box,
which has no GPU.  They monkeypatch ``dream_gnn_amd.ops`` with these functions explicitly; the
product itself never does this and refuses CPU tensors.
"""
import contextlib

import numpy as np
import torch

from oracle import oracle as O


def csr_from_coo(row, col, n_rows, n_cols=0, check_range=False, return_flag=False):
    r, c = row.cpu().numpy(), col.cpu().numpy()
    bad = bool(r.size) and (r.min() < 0 or r.max() >= n_rows or (n_cols > 0 and (c.min() < 0 or c.max() >= n_cols)))
    if bad:  # what the device build leaves behind with the flag set: arrays in bounds, content meaningless
        if check_range:
            raise RuntimeError("csr_from_coo: an id is out of range")
        out = (torch.zeros(n_rows + 1, dtype=torch.int32), torch.zeros(r.size, dtype=torch.int32),
               torch.arange(r.size, dtype=torch.int32))
    else:
        out = tuple(torch.from_numpy(a) for a in O.csr_from_coo(r, c, int(n_rows)))
    return out + (torch.tensor([int(bad)], dtype=torch.int32),) if return_flag else out


def gather_f32(values, perm):
    return values[perm.long()].contiguous()


def spmm_csr_raw(indptr, indices, vals, X, src_scale=None, dst_scale=None, out=None, plan=None):
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    # rows in parallel, each row summed sequentially in fp32: the same values for any thread count
    y = O.spmm_csr(n(indptr), n(indices), n(vals), np.ascontiguousarray(n(X)), n(src_scale), n(dst_scale), threads=O.max_threads())
    y = torch.from_numpy(y)
    if out is not None:
        out.copy_(y)
        return out
    return y


def spmm_csr_raw_f64(indptr, indices, vals, X, src_scale=None, dst_scale=None, out=None, plan=None):
    """The same product evaluated in float64 end to end (torch sparse CSR @ dense, double): the backend of an f64
    evaluation of the whole model (`patched(f64=True)`, the model and its inputs cast to double), against which two fp32
    evaluations can be told apart from each other's rounding."""
    n_dst = indptr.shape[0] - 1
    v = torch.ones(indices.shape[0], dtype=torch.float64) if vals is None else vals.detach().double()
    Xd = X.detach().double()
    if src_scale is not None:
        Xd = Xd * src_scale.detach().double().reshape(-1, 1)
    y = torch.sparse_csr_tensor(indptr.long(), indices.long(), v, (n_dst, Xd.shape[0])) @ Xd
    if dst_scale is not None:
        y = y * dst_scale.detach().double().reshape(-1, 1)
    y = y.to(X.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def gather_concat_raw(src, dst, A, B, out=None, n_src=None, n_dst=None):
    y = torch.from_numpy(O.gather_concat(src.numpy(), dst.numpy(), A.detach().numpy(), B.detach().numpy()))
    if out is not None:
        out.copy_(y)
        return out
    return y


def gather_add_raw(src, dst, A, B, bias=None, out=None, n_src=None, n_dst=None, act=0):
    y = A.detach()[src.long()] + B.detach()[dst.long()]
    if bias is not None:
        y = y + bias.detach()
    if act == 1:
        y = torch.relu(y)
    if out is not None:
        out.copy_(y)
        return out
    return y


def random_subset_mask(E, keep, seed, device):
    return torch.from_numpy(O.random_subset_mask(E, keep, seed))


def random_subset_select(E, keep, seed, device, e_offset=0):
    return torch.from_numpy(O.random_subset_select(E, keep, seed, e_offset).copy())


def random_subset_select_batch(Es, keeps, seeds, device, e_offsets=None):
    offs = [0] * len(Es) if e_offsets is None else list(e_offsets)
    return torch.stack([random_subset_select(E, k, s, device, o) for E, k, s, o in zip(Es, keeps, seeds, offs)])


def keep_mask(desc, E):
    return torch.from_numpy(O.keep_mask(desc.numpy(), E))


def _launch_spmm(dev, indptr, indices, vals, X, src_scale, dst_scale, out, plan, n_dst, n_src, F, ldx,
                 eid=None, keep=None, epi=None):
    if keep is not None:  # edge dropout on the fly: position p takes part iff keep(eid[p])
        m = keep_mask(keep, int(eid.max()) + 1 if eid.numel() else 0)[eid.long()]
        vals = m if vals is None else vals * m
    y = (spmm_csr_raw_f64 if _F64[0] else spmm_csr_raw)(indptr, indices, vals, X, src_scale, dst_scale, out=out)
    if epi is not None:  # the kernels' output epilogue: activation, then the dropout keep mask
        act, slope, mask, mscale = epi
        if act == 1:
            y = torch.nn.functional.leaky_relu_(y, slope)
        if mask is not None:
            y.mul_(mask).mul_(mscale)
    return y


def colsum_rows_(feat_ext, coef, n, R, i0):
    W = feat_ext.shape[1]
    feat_ext[n * R:] = coef @ feat_ext[: n * R].view(n, R, W)[:, i0, :]
    return feat_ext


def colsum_rows_backward_(gf, coef, gs, n, R, i0):
    W = gf.shape[1]
    gf.view(n, R, W)[:, i0, :] += coef.t() @ gs
    return gf


def epilogue_backward(dY, Y, mask, act, slope, mask_scale):
    g = torch.where(Y > 0, dY, dY * slope) if act == 1 else dY.clone()
    return g if mask is None else g * mask * mask_scale


_F64 = [False]


@contextlib.contextmanager
def patched(f64=False):
    """``f64=True``: the SpMM stand-in evaluates in float64 (for a model cast to double)."""
    from dream_gnn_amd import ops

    _F64[0] = bool(f64)

    names = ("csr_from_coo", "gather_f32", "spmm_csr_raw", "_launch_spmm", "_require_device", "build_plan", "FORCE_KERNEL",
             "gather_concat_raw", "gather_add_raw", "random_subset_mask", "random_subset_select", "random_subset_select_batch", "keep_mask", "epilogue_backward", "colsum_rows_", "colsum_rows_backward_")
    saved = {k: getattr(ops, k) for k in names}
    ops.csr_from_coo, ops.gather_f32, ops.spmm_csr_raw = csr_from_coo, gather_f32, (spmm_csr_raw_f64 if f64 else spmm_csr_raw)
    ops._launch_spmm = _launch_spmm
    ops.gather_concat_raw = gather_concat_raw
    ops.gather_add_raw = gather_add_raw
    ops.random_subset_mask = random_subset_mask
    ops.random_subset_select = random_subset_select
    ops.random_subset_select_batch = random_subset_select_batch
    ops.keep_mask = keep_mask
    ops.epilogue_backward = epilogue_backward
    ops.colsum_rows_ = colsum_rows_
    ops.colsum_rows_backward_ = colsum_rows_backward_
    ops._require_device = lambda *ts: next((t.device for t in ts if t is not None), None)
    ops.build_plan = lambda indptr, nnz, chunk=None: None  # launch plans are a device-side concern
    ops.FORCE_KERNEL = "planned"  # the sliced layout is a device-side concern too
    try:
        yield
    finally:
        _F64[0] = False
        for k, v in saved.items():
            setattr(ops, k, v)
