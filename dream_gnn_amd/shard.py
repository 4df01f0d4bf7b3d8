"""Multi-GPU sharding of the SpMM path: one process per GPU, RCCL over xGMI.

The reference is single-device (train.py:459-463); this module exists only for the synthetic
configs large enough to shard (BASELINE cfg 5).  SpMM is linear in the edge set, so any edge
partition gives partial results that sum to the answer.  Two exchange forms are provided:

``rows`` (default)  the edge partition is aligned to destination-row boundaries and balanced
                    by nnz: each rank owns a contiguous row range and all in-edges of those
                    rows.  Partial outputs are disjoint, so the sum degenerates to an
                    **all-gather** of N_dst*F*4/P bytes per rank — no reduction traffic.
                    The backward (A^T dY) uses the same construction on the reversed edges
                    (partitioned by source range), so it is also local-SpMM + all-gather.
``edges``           an arbitrary edge partition: every rank produces a full-height partial Y
                    and the ranks **all-reduce** it (north_star's literal form).  xGMI is
                    point-to-point (7 links per GPU), so a ring all-reduce of N_dst*F*4 bytes is
                    single-link bound; kept for partitions that cannot be row-aligned.

X (source features) is replicated on every rank, as the next layer needs all rows anyway.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all_gather_into_tensor; with the gloo backend (CPU rehearsals of the N>1 path, possibly
    with device tensors) the payload is staged through host memory."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.cpu(), group=group)
        out.copy_(host_out)
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def _direct_exchange(out: torch.Tensor, y_local: torch.Tensor, bounds, rank: int, group=None):
    """All-links exchange of the row blocks: every rank posts its block to each
    of the other ranks and receives theirs straight into its rows of ``out`` — world-1 sends and
    world-1 receives issued as ONE batch, which RCCL runs as a single grouped kernel driving all 7
    xGMI links of the GPU at once (xGMI is point-to-point: a ring all-gather crosses one link per
    step).  Blocks may have different heights (nnz-balanced bounds): no padding, no extra copy of
    the payload.  ``out[bounds[rank]:bounds[rank+1]]`` is filled by a local copy."""
    world = len(bounds) - 1
    out[bounds[rank]:bounds[rank + 1]].copy_(y_local)
    staged = dist.get_backend(group) == "gloo" and y_local.is_cuda  # CPU rehearsal with device tensors
    src = y_local.cpu() if staged else y_local.contiguous()
    recv_bufs, ops_ = {}, []
    for step in range(1, world):  # pairings rotate so that no two ranks target the same peer in a step
        to, frm = (rank + step) % world, (rank - step) % world
        gto = dist.get_global_rank(group, to) if group is not None else to
        gfrm = dist.get_global_rank(group, frm) if group is not None else frm
        if bounds[rank + 1] > bounds[rank]:
            ops_.append(dist.P2POp(dist.isend, src, gto, group))
        if bounds[frm + 1] > bounds[frm]:
            dst_view = out[bounds[frm]:bounds[frm + 1]]
            buf = torch.empty(dst_view.shape, dtype=out.dtype) if staged else dst_view
            recv_bufs[frm] = (buf, dst_view)
            ops_.append(dist.P2POp(dist.irecv, buf, gfrm, group))
    if ops_:
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
    if staged:
        for buf, dst_view in recv_bufs.values():
            dst_view.copy_(buf)
    return out


XGMI_LINKS = 7            # point-to-point links per MI355X in a fully connected 8-GPU node
XGMI_LINK_GBS = 153.0     # per link and direction (prompt / SURVEY §5: 7 x ~153 GB/s per GPU)


def exchange_seconds(recv_bytes: float, world: int, form: str) -> float:
    """The byte model DESIGN §6 states for one row-block exchange in which a rank receives ``recv_bytes``
    in total: ``allgather`` is priced as RCCL's ring (every block crosses ONE link per step, world - 1 steps:
    recv_bytes over one link), ``direct`` as world - 1 concurrent point-to-point transfers, one per link."""
    if world <= 1 or recv_bytes <= 0:
        return 0.0
    links = 1 if form == "allgather" else min(world - 1, XGMI_LINKS)
    return recv_bytes / (links * XGMI_LINK_GBS * 1e9)


def predict_step_seconds(compute_s, recv_bytes, world: int, form: str) -> float:
    """Step time of ``bench.py``'s schedule under that model: the products run back to back on the compute
    stream; each product's exchange is queued on ONE side stream as soon as the product ends and runs behind
    the exchanges queued before it; the step ends when both streams have drained."""
    t_comp = t_comm = 0.0
    for c, b in zip(compute_s, recv_bytes):
        t_comp += c
        t_comm = max(t_comm, t_comp) + exchange_seconds(b, world, form)
    return max(t_comp, t_comm)


def choose_exchange(time_fn, device, world: int, mode: str = "auto", group=None):
    """Pick the row-block exchange form for this job.  ``time_fn(form) -> seconds per step`` runs a few steps
    with that form (a collective: every rank calls it in the same order).  ``mode``: ``"allgather"`` /
    ``"direct"`` force a form; ``"auto"`` times the all-gather (the well-trodden RCCL path) and then the
    all-links form inside ``try`` — a form that raises on ANY rank is dropped on EVERY rank (the failure flag
    is max-reduced before anything else happens), so an unsupported point-to-point path falls back to the
    all-gather instead of killing the job.  Returns ``(form, report)``; ``report`` maps each form tried to its
    max-over-ranks ms per step or to the error text."""
    if world <= 1:
        return None, None
    if mode in ("allgather", "direct"):
        return mode, None
    if dist.get_backend(group) == "gloo":
        device = "cpu"  # CPU rehearsals: the agreement travels as a host tensor
    report, usable = {}, {}
    for form in ("allgather", "direct"):
        err, secs = None, 0.0
        try:
            secs = float(time_fn(form))
        except Exception as exc:  # noqa: BLE001 - whatever the backend throws, the other form is still there
            err = repr(exc)
        flag = torch.tensor([1.0 if err is not None else 0.0, secs], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if float(flag[0]) > 0:
            report[form] = "failed: %s" % (err if err is not None else "on another rank")
        else:
            usable[form] = float(flag[1])
            report[form] = round(float(flag[1]) * 1e3, 3)
    if not usable:
        raise RuntimeError("no row-block exchange form works on this job: %r" % (report,))
    best = min(usable, key=usable.get)  # max-reduced timings: identical on every rank, so is the choice
    return best, report


def balanced_row_bounds(degree: torch.Tensor, parts: int) -> torch.Tensor:
    """Contiguous row ranges with ~equal nnz: int64[parts+1], bounds[0]=0, bounds[-1]=n_rows."""
    n = degree.shape[0]
    csum = torch.cumsum(degree.to(torch.int64), 0)
    total = int(csum[-1]) if n else 0
    targets = torch.arange(1, parts, device=degree.device, dtype=torch.int64) * total // parts
    cuts = torch.searchsorted(csum, targets, right=False) + 1 if n else targets
    cuts = torch.clamp(cuts, 0, n)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=degree.device), cuts,
                        torch.full((1,), n, dtype=torch.int64, device=degree.device)])
    return torch.cummax(bounds, 0)[0]


def choose_row_bounds(degree: torch.Tensor, parts: int, tol: float = 0.01) -> torch.Tensor:
    """Row ranges for ``parts`` ranks: EQUAL row counts when that already balances the edges to within
    ``tol`` of the mean (uniform degrees) — equal blocks make the exchange one unpadded
    ``all_gather_into_tensor`` straight into the result — else the nnz-balanced cut
    (:func:`balanced_row_bounds`)."""
    n = int(degree.shape[0])
    if n % parts == 0 and n > 0:
        per = degree.to(torch.int64).view(parts, n // parts).sum(1)
        mean = float(per.sum()) / parts
        if mean > 0 and float(per.max()) <= (1.0 + tol) * mean:
            return torch.arange(parts + 1, dtype=torch.int64, device=degree.device) * (n // parts)
    return balanced_row_bounds(degree, parts)


class RowShard:
    """This rank's rows [lo, hi) of a relation: local CSR over all in-edges of those rows."""

    def __init__(self, dst: torch.Tensor, src: torch.Tensor, n_dst: int, n_src: int,
                 bounds: torch.Tensor, rank: int, vals: Optional[torch.Tensor] = None):
        self.bounds = [int(b) for b in bounds.tolist()]
        self.rank, self.world = rank, len(self.bounds) - 1
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        self.n_dst, self.n_src = int(n_dst), int(n_src)
        mine = (dst >= self.lo) & (dst < self.hi)
        self.local = ops.CSRGraph((dst[mine] - self.lo).to(torch.int32), src[mine].to(torch.int32),
                                  self.hi - self.lo, n_src, vals=None if vals is None else vals[mine])
        self.max_rows = max(self.bounds[i + 1] - self.bounds[i] for i in range(self.world))

    @property
    def nnz(self) -> int:
        return self.local.nnz

    def spmm_local(self, X, src_scale=None, dst_scale=None, out=None):
        """Rows [lo, hi) of ``diag(dst_scale) A diag(src_scale) X``; scales are full-length."""
        ds = None if dst_scale is None else dst_scale.reshape(-1)[self.lo:self.hi].contiguous()
        return self.local.spmm(X, src_scale, ds, out=out)

    #: how row blocks are exchanged when the caller does not say: "allgather" (one RCCL
    #: all_gather_into_tensor; uneven blocks are padded) or "direct" (batched point-to-point sends to every
    #: peer — all xGMI links at once, no padding).  Same result.  DGMI_EXCHANGE=auto is resolved by the
    #: job (``choose_exchange``: bench.py times both on the real node, guarded, and passes the winner in).
    exchange = {"direct": "direct"}.get(os.environ.get("DGMI_EXCHANGE", ""), "allgather")

    def gather_rows(self, y_local: torch.Tensor, group=None, out: Optional[torch.Tensor] = None,
                    exchange: Optional[str] = None):
        """Assemble the per-rank row blocks into the full (n_dst, F) result on every rank."""
        F = y_local.shape[1]
        exchange = exchange or self.exchange
        if exchange not in ("allgather", "direct"):
            raise ValueError("unknown exchange %r" % (exchange,))
        if exchange == "direct" and self.world > 1:
            if out is None:
                out = torch.empty((self.n_dst, F), dtype=y_local.dtype, device=y_local.device)
            return _direct_exchange(out, y_local, self.bounds, self.rank, group)
        even = all(self.bounds[i + 1] - self.bounds[i] == self.max_rows for i in range(self.world))
        if even:
            if out is None:
                out = torch.empty((self.n_dst, F), dtype=y_local.dtype, device=y_local.device)
            _all_gather_into(out, y_local.contiguous(), group)
            return out
        pad = torch.zeros((self.max_rows, F), dtype=y_local.dtype, device=y_local.device)
        pad[: y_local.shape[0]] = y_local
        buf = torch.empty((self.world * self.max_rows, F), dtype=y_local.dtype, device=y_local.device)
        _all_gather_into(buf, pad, group)
        if out is None:
            out = torch.empty((self.n_dst, F), dtype=y_local.dtype, device=y_local.device)
        for r in range(self.world):
            n = self.bounds[r + 1] - self.bounds[r]
            out[self.bounds[r]:self.bounds[r + 1]] = buf[r * self.max_rows: r * self.max_rows + n]
        return out


class ShardedRelation:
    """A relation sharded over the process group, with autograd: forward = by-destination row
    shard, backward = by-source row shard of the reversed edges; both end in an all-gather."""

    def __init__(self, dst, src, n_dst, n_src, vals=None, group=None, rank=None, world=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_dst, self.n_src = int(n_dst), int(n_src)
        deg_in = torch.bincount(dst.long(), minlength=n_dst)
        deg_out = torch.bincount(src.long(), minlength=n_src)
        self.fwd = RowShard(dst, src, n_dst, n_src, balanced_row_bounds(deg_in, self.world), self.rank, vals)
        self.bwd = RowShard(src, dst, n_src, n_dst, balanced_row_bounds(deg_out, self.world), self.rank, vals)

    def __call__(self, X, src_scale=None, dst_scale=None):
        return _ShardedSpMM.apply(X, self, src_scale, dst_scale)


class _ShardedSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, rel: ShardedRelation, src_scale, dst_scale):
        ctx.rel = rel
        ctx.save_for_backward(src_scale, dst_scale)
        return rel.fwd.gather_rows(rel.fwd.spmm_local(X, src_scale, dst_scale), rel.group)

    @staticmethod
    def backward(ctx, dY):
        src_scale, dst_scale = ctx.saved_tensors
        rel = ctx.rel
        # dX = diag(src_scale) A^T diag(dst_scale) dY: rows = source nodes of this rank's range
        dX = rel.bwd.gather_rows(rel.bwd.spmm_local(dY.contiguous(), dst_scale, src_scale), rel.group)
        return dX, None, None, None


class EdgeShard:
    """Arbitrary edge partition + sum all-reduce of the full-height partial result."""

    def __init__(self, dst, src, n_dst, n_src, vals=None, group=None):
        self.group = group
        self.local = ops.CSRGraph(dst.to(torch.int32), src.to(torch.int32), n_dst, n_src, vals=vals)

    def spmm(self, X, src_scale=None, dst_scale=None):
        # dst_scale is linear: apply it after the reduce (once), src_scale inside the kernel
        y = self.local.spmm(X, src_scale, None)
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        if dst_scale is not None:
            y = y * dst_scale.reshape(-1, 1)
        return y
