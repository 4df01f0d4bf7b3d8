"""The N>1 path on CPU: world_size-2 gloo, local SpMM served by the oracle-backed test backend.
Checks both exchange forms (row-aligned all-gather with autograd; arbitrary edge partition with
all-reduce) against the single-process oracle, on a skewed graph so the nnz-balanced row
bounds are uneven (padded all-gather path) and on an even one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(skew):
    rng = np.random.default_rng(5)
    n_dst, n_src, E, F = (64, 48, 3000, 20) if skew else (64, 64, 64 * 16, 20)
    if skew:
        p = 1.0 / np.arange(1, n_dst + 1) ** 1.1
        dst = rng.choice(n_dst, size=E, p=p / p.sum())
        src = rng.integers(0, n_src, E)
    else:  # every row and every column has exactly 16 edges -> even bounds
        dst = np.repeat(np.arange(n_dst), 16)
        src = (dst * 7 + np.tile(np.arange(16), n_dst) * 5) % n_src
    val = rng.standard_normal(E if skew else dst.size).astype(np.float32)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    dY = rng.standard_normal((n_dst, F)).astype(np.float32)
    ss = rng.uniform(0.2, 1, n_src).astype(np.float32)
    ds = rng.uniform(0.2, 1, n_dst).astype(np.float32)
    return dst.astype(np.int64), src.astype(np.int64), val, X, dY, ss, ds, n_dst, n_src


def _worker(rank, world, port, skew, q, exchange="allgather"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_backend
        from dream_gnn_amd import shard
        from oracle import oracle as O

        shard.RowShard.exchange = exchange
        dst, src, val, X, dY, ss, ds, n_dst, n_src = _graph(skew)
        t = torch.from_numpy
        with _cpu_backend.patched():
            rel = shard.ShardedRelation(t(dst), t(src), n_dst, n_src, vals=t(val))
            x = t(X).requires_grad_(True)
            y = rel(x, t(ss), t(ds))
            y.backward(t(dY))
            # arbitrary (strided) edge partition + all-reduce
            mine = np.arange(dst.size) % world == rank
            es = shard.EdgeShard(t(dst[mine]), t(src[mine]), n_dst, n_src, vals=t(val[mine]))
            y2 = es.spmm(t(X), t(ss), t(ds))
        ip, ix, eid = O.csr_from_coo(dst, src, n_dst)
        ref = O.spmm_csr(ip, ix, val[eid], X, ss, ds, acc="f64")
        tp, ti, te = O.csr_from_coo(src, dst, n_src)
        ref_dx = O.spmm_csr(tp, ti, val[te], dY, ds, ss, acc="f64")
        scale = np.abs(ref).max()
        ok = (np.abs(y.detach().numpy() - ref).max() <= 1e-5 * scale
              and np.abs(y2.numpy() - ref).max() <= 1e-5 * scale
              and np.abs(x.grad.numpy() - ref_dx).max() <= 1e-5 * np.abs(ref_dx).max())
        bounds = rel.fwd.bounds
        covered = sum(dist_nnz for dist_nnz in [rel.fwd.nnz])
        tot = torch.tensor([covered])
        dist.all_reduce(tot)
        ok = ok and int(tot) == dst.size and bounds[0] == 0 and bounds[-1] == n_dst
        q.put((rank, bool(ok), bounds, rel.fwd.nnz))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "direct"])
@pytest.mark.parametrize("skew", [True, False])
def test_sharded_spmm_world2(oracle, skew, exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, skew, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    nnz = sorted(r[3] for r in res)
    if skew:  # nnz-balanced, not row-balanced
        b = res[0][2]
        assert b[1] != 32 and nnz[1] - nnz[0] <= 0.25 * sum(nnz)
    else:
        assert res[0][2] == [0, 32, 64] and nnz[0] == nnz[1]


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dream_gnn_amd import shard

        bounds = [0, 5, 5, 12, 13]  # uneven blocks, rank 1 owns nothing
        F = 3
        full = torch.arange(13 * F, dtype=torch.float32).view(13, F)
        mine = full[bounds[rank]:bounds[rank + 1]].clone()
        out = torch.full((13, F), -1.0)
        shard._direct_exchange(out, mine, bounds, rank)
        q.put((rank, bool(torch.equal(out, full))))
    finally:
        dist.destroy_process_group()


def test_direct_exchange_indexing_world4():
    """The all-links exchange (every rank posts its block to every peer, receives into its rows of
    the result): uneven blocks and an empty one, no padding — indexing only, on gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _auto_worker(rank, world, port, q, break_direct):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time

        import _cpu_backend
        from dream_gnn_amd import shard

        dst, src, val, X, dY, ss, ds, n_dst, n_src = _graph(True)
        t = torch.from_numpy
        if break_direct:  # what an unsupported point-to-point path looks like: the call raises
            def broken(*a, **k):
                raise RuntimeError("batch_isend_irecv: not supported on this transport")
            shard._direct_exchange = broken
        with _cpu_backend.patched():
            deg = torch.bincount(t(dst), minlength=n_dst)
            sh = shard.RowShard(t(dst), t(src), n_dst, n_src, shard.balanced_row_bounds(deg, world), rank, vals=t(val))
            y_local = sh.spmm_local(t(X), t(ss), t(ds))

            def time_fn(form):
                t0 = time.perf_counter()
                sh.gather_rows(y_local, exchange=form)
                return time.perf_counter() - t0

            form, report = shard.choose_exchange(time_fn, "cpu", world, mode="auto")
            y = sh.gather_rows(y_local, exchange=form)
            forced, none = shard.choose_exchange(time_fn, "cpu", world, mode="allgather")
        q.put((rank, form, report, y.numpy().copy(), forced, none))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("break_direct", [True, False])
def test_exchange_auto_falls_back_when_the_all_links_form_raises(oracle, break_direct):
    """bench.py's default (DGMI_EXCHANGE=auto -> shard.choose_exchange): both forms are tried, a form that raises is
    dropped on EVERY rank (no rank is left inside a collective), all ranks agree on the survivor, and the result
    assembled with it is the full product."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_auto_worker, args=(r, 2, port, q, break_direct)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dst, src, val, X, dY, ss, ds, n_dst, n_src = _graph(True)
    ip, ix, eid = oracle.csr_from_coo(dst, src, n_dst)
    ref = oracle.spmm_csr(ip, ix, val[eid], X, ss, ds, acc="f64")
    forms = {r[1] for r in res}
    assert len(forms) == 1 and forms <= {"allgather", "direct"}
    for _, form, report, y, forced, none in res:
        assert set(report) == {"allgather", "direct"} and forced == "allgather" and none is None
        assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
        if break_direct:
            assert form == "allgather" and str(report["direct"]).startswith("failed") and isinstance(report["allgather"], float)
        else:
            assert isinstance(report["direct"], float)


def test_exchange_byte_model():
    """DESIGN 6's prediction model as bench.py applies it (host arithmetic)."""
    from dream_gnn_amd import shard as S

    assert S.exchange_seconds(153e9, 8, "allgather") == pytest.approx(1.0) and S.exchange_seconds(153e9, 8, "direct") == pytest.approx(1 / 7)
    assert S.exchange_seconds(153e9, 2, "direct") == pytest.approx(1.0) and S.exchange_seconds(1e9, 1, "direct") == 0.0
    # two products of 1 s each; the first exchange (0.5 s) hides behind the second product, the last one is exposed
    assert S.predict_step_seconds([1.0, 1.0], [0.5 * 153e9, 0.25 * 153e9], 8, "allgather") == pytest.approx(2.25)
    # an exchange longer than the next product queues the following one behind it
    assert S.predict_step_seconds([1.0, 1.0], [2 * 153e9, 153e9], 8, "allgather") == pytest.approx(4.0)
    assert S.predict_step_seconds([1.0, 1.0], [0.0, 0.0], 1, "direct") == pytest.approx(2.0)


def test_balanced_row_bounds_properties():
    from dream_gnn_amd.shard import balanced_row_bounds, choose_row_bounds

    uniform = torch.full((800,), 100) + (torch.arange(800) % 3)        # equal rows balance the edges: even blocks
    assert choose_row_bounds(uniform, 8).tolist() == list(range(0, 801, 100))
    skewed = torch.cat([torch.full((100,), 1000), torch.ones(700, dtype=torch.int64)])
    b = choose_row_bounds(skewed, 8).tolist()                           # not balanced by rows: nnz cut
    assert b == balanced_row_bounds(skewed, 8).tolist() and b != list(range(0, 801, 100))
    assert choose_row_bounds(torch.ones(10, dtype=torch.int64), 4).tolist() == balanced_row_bounds(torch.ones(10, dtype=torch.int64), 4).tolist()

    deg = torch.tensor([0, 0, 10, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1])
    b = balanced_row_bounds(deg, 2).tolist()
    assert b[0] == 0 and b[-1] == deg.numel() and b == sorted(b)
    assert balanced_row_bounds(torch.zeros(5, dtype=torch.int64), 4).tolist()[-1] == 5
    b8 = balanced_row_bounds(torch.ones(80, dtype=torch.int64), 8).tolist()
    assert b8 == list(range(0, 81, 10))
