"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances (north_star): CSR indexing bit-exact; fp32 embeddings within 1e-5 relative
(see _check_close for what "relative" is measured against).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _rand_graph(rng, n_dst, n_src, E, skew=False, empty_rows=True):
    if skew:
        p = 1.0 / np.arange(1, n_dst + 1) ** 1.2
        p /= p.sum()
        dst = rng.choice(n_dst, size=E, p=p).astype(np.int32)
    else:
        dst = rng.integers(0, n_dst, size=E, dtype=np.int32)
    if empty_rows and n_dst > 4:
        dst[dst == 1] = 0
        dst[dst == n_dst - 1] = n_dst - 2
    src = rng.integers(0, n_src, size=E, dtype=np.int32)
    return dst, src


def _check_close(y_hip, y32, y64, yabs):
    """(a) forward-error bound of any fp32 summation order, elementwise, against the f64 oracle;
    (b) 1e-5 relative to the output's magnitude, against the f64 oracle;
    (c) no further from the sequential-fp32 oracle than that oracle's own rounding + 1e-5 rel
        (a 60k-edge row summed sequentially in fp32 is itself ~6e-5 off)."""
    y_hip = y_hip.astype(np.float64)
    err = np.abs(y_hip - y64)
    bound = RTOL * yabs + 1e-30
    assert np.all(err <= bound), "max err/bound = %g" % float((err / bound).max())
    if y64.size == 0:
        return
    scale = max(float(np.abs(y64).max()), 1e-30)
    assert float(err.max()) <= RTOL * scale, "rel err %g" % (float(err.max()) / scale)
    assert float(np.abs(y_hip - y32).max()) <= float(np.abs(y32 - y64).max()) + RTOL * scale


def _plans(ops, g):
    """launch forms to cover: a wave per row, the default plan, and a plan with 16-edge chunks
    (forces multi-chunk rows + the reduce pass on small graphs)."""
    return [None, g.plan, ops.build_plan(g.indptr, g.nnz, chunk=16)]


@pytest.mark.parametrize("F", [1, 3, 4, 8, 32, 64, 100, 127, 128, 256, 341, 344, 768])
@pytest.mark.parametrize("mode", ["copy_u", "copy_u_scaled", "u_mul_e", "all"])
def test_spmm_vs_oracle(oracle, dev, F, mode):
    from dream_gnn_amd import ops

    rng = np.random.default_rng(1000 + F)
    n_dst, n_src, E = 257, 301, 9000
    dst, src = _rand_graph(rng, n_dst, n_src, E)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    vals = rng.standard_normal(E).astype(np.float32) if mode in ("u_mul_e", "all") else None
    ss = rng.uniform(0.1, 1.0, n_src).astype(np.float32) if mode in ("copy_u_scaled", "all") else None
    ds = rng.uniform(0.1, 1.0, n_dst).astype(np.float32) if mode in ("copy_u_scaled", "all") else None

    indptr, indices, eid = oracle.csr_from_coo(dst, src, n_dst)
    v_csr = None if vals is None else vals[eid]
    y32 = oracle.spmm_csr(indptr, indices, v_csr, X, ss, ds)
    y64 = oracle.spmm_csr(indptr, indices, v_csr, X, ss, ds, acc="f64")
    yabs = oracle.spmm_csr(indptr, indices, v_csr, X, ss, ds, acc="abs")

    t = lambda a, dt=None: None if a is None else torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(dst), t(src), n_dst, n_src, vals=t(vals))
    # CSR indexing: bit-exact
    assert np.array_equal(g.indptr.cpu().numpy(), indptr)
    assert np.array_equal(g.indices.cpu().numpy(), indices)
    assert np.array_equal(g.eid.cpu().numpy(), eid)
    deg = np.diff(indptr)
    for plan in _plans(ops, g):
        y = ops.spmm_csr_raw(g.indptr, g.indices, g.vals, t(X), t(ss), t(ds), plan=plan)
        torch.cuda.synchronize()
        _check_close(y.cpu().numpy(), y32, y64, yabs)
        assert np.all(y.cpu().numpy()[deg == 0] == 0)  # empty rows are exactly zero


@pytest.mark.parametrize("case", ["E0", "one_row", "one_edge", "long_row", "skew", "dups"])
def test_spmm_edge_cases(oracle, dev, case):
    from dream_gnn_amd import ops

    rng = np.random.default_rng(7)
    F = 128
    if case == "E0":
        n_dst, n_src = 5, 4
        dst = np.zeros(0, np.int32); src = np.zeros(0, np.int32)
    elif case == "one_row":
        n_dst, n_src = 1, 50
        dst = np.zeros(333, np.int32); src = rng.integers(0, n_src, 333, dtype=np.int32)
    elif case == "one_edge":
        n_dst, n_src = 3, 3
        dst = np.array([2], np.int32); src = np.array([1], np.int32)
    elif case == "long_row":
        n_dst, n_src = 40, 2000
        dst = np.concatenate([np.full(20000, 17, np.int32), rng.integers(0, n_dst, 500, dtype=np.int32)])
        src = rng.integers(0, n_src, dst.shape[0], dtype=np.int32)
    elif case == "skew":
        n_dst, n_src = 3000, 1000
        dst, src = _rand_graph(rng, n_dst, n_src, 200000, skew=True)
    else:  # duplicates: multigraph semantics, every copy counts
        n_dst, n_src = 10, 10
        dst = np.array([3, 3, 3, 3, 5, 5], np.int32); src = np.array([2, 2, 2, 7, 1, 1], np.int32)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    indptr, indices, eid = oracle.csr_from_coo(dst, src, n_dst)
    y32 = oracle.spmm_csr(indptr, indices, None, X)
    y64 = oracle.spmm_csr(indptr, indices, None, X, acc="f64")
    yabs = oracle.spmm_csr(indptr, indices, None, X, acc="abs")
    g = ops.CSRGraph(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n_dst, n_src)
    assert np.array_equal(g.indptr.cpu().numpy(), indptr)
    assert np.array_equal(g.indices.cpu().numpy(), indices)
    assert np.array_equal(g.eid.cpu().numpy(), eid)
    for plan in _plans(ops, g):
        y = ops.spmm_csr_raw(g.indptr, g.indices, None, torch.from_numpy(X).to(dev), plan=plan)
        torch.cuda.synchronize()
        _check_close(y.cpu().numpy(), y32, y64, yabs)
    # the plan itself: every edge in exactly one item, chunks <= chunk, long rows listed once
    n_items, n_long, n_slots, chunk = g.plan.header()[:4]
    want_items = int(np.maximum(1, -(-deg // chunk)).sum()) if (deg := np.diff(indptr)).size else 0
    assert n_items == want_items and n_long == int((deg > chunk).sum())
    assert n_slots == int((-(-deg // chunk))[deg > chunk].sum())


def test_strided_input_and_determinism(oracle, dev):
    from dream_gnn_amd import ops

    rng = np.random.default_rng(3)
    n_dst, n_src, E, F = 500, 400, 30000, 128
    dst, src = _rand_graph(rng, n_dst, n_src, E)
    Xbig = rng.standard_normal((n_src, 2 * F)).astype(np.float32)
    g = ops.CSRGraph(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n_dst, n_src)
    xb = torch.from_numpy(Xbig).to(dev)
    y1 = ops.spmm_csr_raw(g.indptr, g.indices, None, xb[:, F:])  # row-strided view, ldx = 2F
    y2 = ops.spmm_csr_raw(g.indptr, g.indices, None, xb[:, F:].contiguous())
    y3 = ops.spmm_csr_raw(g.indptr, g.indices, None, xb[:, F:])
    assert torch.equal(y1, y2) and torch.equal(y1, y3)  # bitwise reproducible
    p16 = ops.build_plan(g.indptr, g.nnz, chunk=16)
    z1 = ops.spmm_csr_raw(g.indptr, g.indices, None, xb[:, F:], plan=p16)
    z2 = ops.spmm_csr_raw(g.indptr, g.indices, None, xb[:, F:], plan=p16)
    assert torch.equal(z1, z2)  # chunk partials are added in chunk order: reproducible too
    indptr, indices, _ = oracle.csr_from_coo(dst, src, n_dst)
    y32 = oracle.spmm_csr(indptr, indices, None, np.ascontiguousarray(Xbig[:, F:]))
    assert np.abs(y1.cpu().numpy() - y32).max() <= RTOL * np.abs(y32).max()


def test_autograd_is_transpose_spmm(oracle, dev):
    from dream_gnn_amd import ops

    rng = np.random.default_rng(11)
    n_dst, n_src, E, F = 300, 200, 5000, 128
    dst, src = _rand_graph(rng, n_dst, n_src, E)
    vals = rng.standard_normal(E).astype(np.float32)
    ss = rng.uniform(0.1, 1, n_src).astype(np.float32)
    ds = rng.uniform(0.1, 1, n_dst).astype(np.float32)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    dY = rng.standard_normal((n_dst, F)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(dst), t(src), n_dst, n_src, vals=t(vals))
    x = t(X).requires_grad_(True)
    y = ops.spmm_csr(g, x, t(ss), t(ds))
    y.backward(t(dY))
    # oracle: dX = diag(ss) A^T diag(ds) dY on the reversed edges
    indptr_t, indices_t, eid_t = oracle.csr_from_coo(src, dst, n_src)
    ref = oracle.spmm_csr(indptr_t, indices_t, vals[eid_t], dY, ds, ss, acc="f64")
    refabs = oracle.spmm_csr(indptr_t, indices_t, vals[eid_t], dY, ds, ss, acc="abs")
    err = np.abs(x.grad.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= RTOL * refabs + 1e-30)


def test_out_of_range_row_is_reported(dev):
    from dream_gnn_amd import ops

    row = torch.tensor([0, 5, 1], dtype=torch.int32, device=dev)
    col = torch.tensor([0, 0, 0], dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        ops.csr_from_coo(row, col, 3, check_range=True)
    ok_row = torch.tensor([0, 2, 1], dtype=torch.int32, device=dev)
    bad_col = torch.tensor([0, 7, 0], dtype=torch.int32, device=dev)
    ops.csr_from_coo(ok_row, bad_col, 3, check_range=True)  # columns unchecked without n_cols
    with pytest.raises(RuntimeError):
        ops.csr_from_coo(ok_row, bad_col, 3, n_cols=4, check_range=True)
    with pytest.raises(RuntimeError):
        ops.CSRGraph(ok_row, bad_col, 3, 4)


@pytest.mark.parametrize("bad", [-1, 3 + 2 ** 10, 3 + 2 ** 20, 2 ** 31 - 1])
@pytest.mark.parametrize("side", ["dst", "src"])
def test_csrgraph_rejects_bad_ids_before_building_anything(dev, bad, side):
    """a negative or aliasing id must surface as RuntimeError from CSRGraph itself — the
    range flag is read before the launch plan is derived from the (then meaningless) indptr — and
    the device must still be healthy afterwards."""
    from dream_gnn_amd import ops

    n_dst, n_src, E = 3, 5, 4000
    g = torch.Generator().manual_seed(1)
    dst = torch.randint(0, n_dst, (E,), generator=g, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=g, dtype=torch.int32)
    (dst if side == "dst" else src)[E // 2] = bad
    with pytest.raises(RuntimeError, match="out of range"):
        ops.CSRGraph(dst.to(dev), src.to(dev), n_dst, n_src)
    with pytest.raises(RuntimeError, match="out of range"):
        ops.EdgePairs(src.to(dev), dst.to(dev), n_src, n_dst)
    torch.cuda.synchronize()
    ok = ops.CSRGraph(torch.zeros(4, dtype=torch.int32, device=dev), torch.arange(4, dtype=torch.int32, device=dev), 1, 4)
    y = ok.spmm(torch.ones(4, 8, device=dev))
    assert torch.equal(y.cpu(), torch.full((1, 8), 4.0))


@pytest.mark.parametrize("Fa,Fb", [(128, 128), (16, 16), (8, 24), (5, 7), (341, 3)])
def test_gather_concat_bit_exact_and_backward(oracle, dev, Fa, Fb):
    """(f2) decoder apply_edges(udf_u_mul_e): forward is pure copies -> bit-identical to the
    oracle; backward (copy_e -> sum) is the SpMM kernel over edge-id CSRs -> 1e-5 relative."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(Fa * 1000 + Fb)
    n_a, n_b, E = 61, 47, 5000
    src = rng.integers(0, n_a, E, dtype=np.int32)
    dst = rng.integers(0, n_b, E, dtype=np.int32)
    src[src == 3] = 4  # node 3 has no edge -> zero gradient row
    A = rng.standard_normal((n_a, Fa)).astype(np.float32)
    B = rng.standard_normal((n_b, Fb)).astype(np.float32)
    dO = rng.standard_normal((E, Fa + Fb)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs = ops.EdgePairs(t(src), t(dst), n_a, n_b)
    a, b = t(A).requires_grad_(True), t(B).requires_grad_(True)
    out = ops.gather_concat(pairs, a, b)
    out.backward(t(dO))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.gather_concat(src, dst, A, B))
    for grad, key, n, cols in ((a.grad, src, n_a, slice(0, Fa)), (b.grad, dst, n_b, slice(Fa, Fa + Fb))):
        ip, ix, _ = oracle.csr_from_coo(key, np.arange(E, dtype=np.int32), n)
        ref = oracle.spmm_csr(ip, ix, None, np.ascontiguousarray(dO[:, cols]), acc="f64")
        assert np.abs(grad.cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)
    assert np.all(a.grad.cpu().numpy()[3] == 0)


def test_gather_concat_empty_and_range_check(dev):
    from dream_gnn_amd import ops

    z = torch.zeros(0, dtype=torch.int32, device=dev)
    out = ops.gather_concat_raw(z, z, torch.randn(3, 8, device=dev), torch.randn(2, 8, device=dev))
    assert out.shape == (0, 16)
    with pytest.raises(RuntimeError):
        ops.EdgePairs(torch.tensor([0, 9], dtype=torch.int32, device=dev),
                      torch.tensor([0, 0], dtype=torch.int32, device=dev), 3, 3)


@pytest.mark.parametrize("F", [4, 32, 64, 128, 256, 344])
@pytest.mark.parametrize("mode", ["copy_u", "all"])
@pytest.mark.parametrize("n_slices", [8, 3])
def test_xcd_sliced_spmm_vs_oracle(oracle, dev, F, mode, n_slices):
    """The XCD-local path: sliced CSR layout bit-exact vs its numpy restatement, product within
    1e-5 of the f64 oracle (its in-row order is slice by slice, so it is compared against f64,
    not against the plain-CSR fp32 order)."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(F + n_slices)
    n_dst, n_src, E = 203, 157, 7000
    dst, src = _rand_graph(rng, n_dst, n_src, E)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    vals = rng.standard_normal(E).astype(np.float32) if mode == "all" else None
    ss = rng.uniform(0.1, 1.0, n_src).astype(np.float32) if mode == "all" else None
    ds = rng.uniform(0.1, 1.0, n_dst).astype(np.float32) if mode == "all" else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    sl = ops.SlicedCSR(t(dst), t(src), n_dst, n_src, vals=t(vals), n_slices=n_slices)
    segptr, indices, eid = oracle.csr_sliced_from_coo(dst, src, n_dst, n_src, n_slices)
    assert np.array_equal(sl.segptr.cpu().numpy(), segptr)
    assert np.array_equal(sl.indices.cpu().numpy(), indices)
    assert np.array_equal(sl.eid.cpu().numpy(), eid)
    y = sl.spmm(t(X), t(ss), t(ds)).cpu().numpy()
    ip, ix, e0 = oracle.csr_from_coo(dst, src, n_dst)
    v0 = None if vals is None else vals[e0]
    y64 = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="f64")
    yabs = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="abs")
    assert np.all(np.abs(y - y64) <= RTOL * yabs + 1e-30)
    assert np.abs(y - y64).max() <= RTOL * np.abs(y64).max()
    assert np.all(y[np.diff(ip) == 0] == 0)
    assert np.array_equal(y, sl.spmm(t(X), t(ss), t(ds)).cpu().numpy())  # reproducible


@pytest.mark.parametrize("F", [128, 256, 344, 64])
@pytest.mark.parametrize("mode", ["copy_u", "all"])
def test_xcd_sliced_column_passes(oracle, dev, F, mode):
    """With > 65536 sources at F = 128 (and >= 32768 destination rows) an XCD's slice of X exceeds its 4 MiB L2
    and the launcher sweeps the columns in two half-width passes (16-lane groups; F = 256: 64 -> 32 lanes; F = 344: a ragged last tile;
    F = 64: slice still under 4 MiB, one pass).  The per-element sum order does not depend on the column
    tiling: bit-identical to the single full-width pass (`column_passes = 1`), and within 1e-5 of f64."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(F)
    n_dst, n_src, E = 33001, 70001, 90000
    dst, src = _rand_graph(rng, n_dst, n_src, E)
    dst[:20000] = rng.integers(0, 300, 20000)  # some rows of ~70 edges among mostly short ones
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    vals = rng.standard_normal(E).astype(np.float32) if mode == "all" else None
    ss = rng.uniform(0.1, 1.0, n_src).astype(np.float32) if mode == "all" else None
    ds = rng.uniform(0.1, 1.0, n_dst).astype(np.float32) if mode == "all" else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    sl = ops.SlicedCSR(t(dst), t(src), n_dst, n_src, vals=t(vals))
    y_auto = sl.spmm(t(X), t(ss), t(ds))
    y_full = sl.spmm(t(X), t(ss), t(ds), full_width=True)
    assert torch.equal(y_auto, y_full)
    ip, ix, e0 = oracle.csr_from_coo(dst, src, n_dst)
    v0 = None if vals is None else vals[e0]
    y64 = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="f64")
    yabs = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="abs")
    y = y_auto.cpu().numpy()
    assert np.all(np.abs(y - y64) <= RTOL * yabs + 1e-30)
    assert np.abs(y - y64).max() <= RTOL * np.abs(y64).max()


@pytest.mark.parametrize("n_slices", [8, 3, 1])
@pytest.mark.parametrize("shape", [(203, 157, 7000), (5, 3, 0), (9, 5, 5300), (4000, 70001, 90000)])
def test_sliced_layout_from_csr_is_bit_identical(oracle, dev, shape, n_slices):
    """(f1) `dgmi_csr_sliced_from_csr_i32`: the sliced layout derived from the CSR by one stable partition
    pass == the layout sorted from the COO list == its numpy restatement, bit for bit (segptr, indices, eid)."""
    from dream_gnn_amd import ops

    n_dst, n_src, E = shape
    rng = np.random.default_rng(n_dst + n_slices)
    dst, src = _rand_graph(rng, n_dst, n_src, E) if E else (np.zeros(0, np.int32), np.zeros(0, np.int32))
    if E:
        dst[: E // 3] = 2  # one long row
    d, s_ = torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev)
    indptr, indices, eid, _ = ops.csr_from_coo(d, s_, n_dst, n_src, return_flag=True)
    a = ops.SlicedCSR(d, s_, n_dst, n_src, n_slices=n_slices)
    b = ops.SlicedCSR.from_csr(indptr, indices, eid, n_dst, n_src, n_slices=n_slices)
    assert torch.equal(a.segptr, b.segptr) and torch.equal(a.indices, b.indices) and torch.equal(a.eid, b.eid)
    assert int(b.range_flag) == 0
    segptr, idx, e = oracle.csr_sliced_from_coo(dst, src, n_dst, n_src, n_slices)
    assert np.array_equal(b.segptr.cpu().numpy(), segptr) and np.array_equal(b.indices.cpu().numpy(), idx)
    assert np.array_equal(b.eid.cpu().numpy(), e)


@pytest.mark.parametrize("shape", [
    (300, 200, 8191), (300, 200, 8192), (300, 200, 8193), (513, 90, 16385),      # tile edges, 1 and 2 digit passes
    (300_000, 1000, 70_001), (2_000_000, 10, 300_000),                          # 3 digit passes; empty trailing slices
    (1_000_000, 700, 50_000, "few_rows"), (40, 40, 200_000, "one_row"), (7, 9, 1)])
def test_record_sort_builds_every_layout_bit_exact(oracle, dev, shape):
    """(f1) the hand-written record radix sort (`csrc/dgmi_sort.hip`) behind all three builders: CSR, sliced from COO,
    sliced from the CSR == the oracle's stable sorts, bit for bit — at tile edges (8192 records per tile), at 1 / 2 / 3
    digit passes, with every edge in one row (one digit bucket takes everything), and with long runs of empty keys
    (edges confined to a few rows of a 10^6-row range; empty trailing slices — the runs the boundary pass fills
    wave-wide)."""
    from dream_gnn_amd import ops

    n_dst, n_src, E = shape[:3]
    kind = shape[3] if len(shape) > 3 else ""
    rng = np.random.default_rng(E)
    dst = rng.integers(0, n_dst, E).astype(np.int32)
    src = rng.integers(0, n_src, E).astype(np.int32)
    if kind == "few_rows":
        dst = (rng.integers(0, 5, E) * 199_999 + 17).astype(np.int32)
    if kind == "one_row":
        dst[:] = 33
    d, s_ = torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev)
    indptr, indices, eid, flag = ops.csr_from_coo(d, s_, n_dst, n_src, return_flag=True)
    ip, ix, e = oracle.csr_from_coo(dst, src, n_dst)
    assert int(flag) == 0
    assert np.array_equal(indptr.cpu().numpy(), ip) and np.array_equal(indices.cpu().numpy(), ix)
    assert np.array_equal(eid.cpu().numpy(), e)
    for n_slices in (8, 64) if n_dst * 64 < 2**26 else (8,):
        a = ops.SlicedCSR(d, s_, n_dst, n_src, n_slices=n_slices)
        b = ops.SlicedCSR.from_csr(indptr, indices, eid, n_dst, n_src, n_slices=n_slices)
        segptr, idx, e2 = oracle.csr_sliced_from_coo(dst, src, n_dst, n_src, n_slices)
        for got in (a, b):
            assert int(got.range_flag) == 0
            assert np.array_equal(got.segptr.cpu().numpy(), segptr)
            assert np.array_equal(got.indices.cpu().numpy(), idx) and np.array_equal(got.eid.cpu().numpy(), e2)


def test_xcd_sliced_edge_cases(oracle, dev):
    from dream_gnn_amd import ops

    z = torch.zeros(0, dtype=torch.int32, device=dev)
    sl = ops.SlicedCSR(z, z, 5, 3)  # no edges, fewer sources than slices
    assert torch.equal(sl.spmm(torch.randn(3, 8, device=dev)), torch.zeros(5, 8, device=dev))
    rng = np.random.default_rng(1)
    dst = np.concatenate([np.full(5000, 2, np.int32), rng.integers(0, 9, 300, dtype=np.int32)])  # one long row
    src = rng.integers(0, 5, dst.size, dtype=np.int32)
    X = rng.standard_normal((5, 128)).astype(np.float32)
    sl = ops.SlicedCSR(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), 9, 5)
    ip, ix, _ = oracle.csr_from_coo(dst, src, 9)
    y64 = oracle.spmm_csr(ip, ix, None, X, acc="f64")
    y = sl.spmm(torch.from_numpy(X).to(dev)).cpu().numpy()
    assert np.abs(y - y64).max() <= 1e-5 * np.abs(y64).max()


@pytest.mark.skipif(bool(__import__("os").environ.get("DGMI_FORCE_KERNEL")), reason="kernel choice is forced")
def test_csrgraph_picks_sliced_only_when_profitable(dev):
    from dream_gnn_amd import ops

    gen = torch.Generator(device=dev).manual_seed(0)
    n_dst, n_src, E = 20_000, 40_000, 2_000_000  # table 40k x 128 x 4 = 20 MB, avg degree 100
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
    g = ops.CSRGraph(dst, src, n_dst, n_src)
    assert g.regular
    X = torch.randn(n_src, 128, device=dev)
    y = g.spmm(X)
    assert g._sliced is not None  # XCD-local path taken
    y_plain = ops.spmm_csr_raw(g.indptr, g.indices, None, X)
    assert float((y - y_plain).abs().max()) <= 1e-5 * float(y_plain.abs().max())
    W = torch.randn(n_dst, 128, device=dev)
    dx = g.spmm_t(W)
    it, ix, _, _ = g.transposed()
    dx_plain = ops.spmm_csr_raw(it, ix, None, W)
    assert float((dx - dx_plain).abs().max()) <= 1e-5 * float(dx_plain.abs().max())
    g.spmm(torch.randn(n_src, 16, device=dev))  # 2.5 MB table: stays on the planned kernel
    small = ops.CSRGraph(dst[:1000] % 50, src[:1000] % 60, 50, 60)
    small.spmm(torch.randn(60, 128, device=dev))
    assert small._sliced is None
    # the fitted rule (tools/kernel_choice_sweep.py): by average degree and table size
    assert g._use_sliced(128, 20_000, 40_000, True) and not g._use_sliced(16, 20_000, 40_000, True)       # 20 MB / 2.5 MB at degree 100
    assert g._use_sliced(128, 20_000, 1_500_000, True) and not g._use_sliced(128, 20_000, 1_700_000, True)  # degree 100: up to 800 MB
    assert not g._use_sliced(128, 100_000, 40_000, True)   # degree 20: a (row, slice) segment of 2-3 edges is all overhead
    assert g._use_sliced(128, 50_000, 100_000, True) and not g._use_sliced(128, 50_000, 300_000, True)    # degree 40: only ~L2-sized slices
    assert not g._use_sliced(128, 20_000, 40_000, False) and not g._use_sliced(126, 20_000, 40_000, True)
    # a power-law graph is not "regular": never the plain sliced form — its rows are cut into virtual rows first
    p = 1.0 / torch.arange(1, n_dst + 1, device=dev, dtype=torch.float64) ** 1.2
    skew = ops.CSRGraph(torch.multinomial(p / p.sum(), E, replacement=True, generator=gen).to(torch.int32), src, n_dst, n_src)
    ys = skew.spmm(X)
    assert not skew.regular and skew._sliced is None and skew._S.split is not None
    ys_plain = ops.spmm_csr_raw(skew.indptr, skew.indices, None, X, plan=skew.plan)
    assert float((ys - ys_plain).abs().max()) <= 1e-5 * float(ys_plain.abs().max())
    # (r4) an UNCHECKED build (what graph.similarity_graph / feature_similarity_graph hand over: trusted by construction)
    # learns whether it is regular at its first LARGE product — until round 4 it never took the XCD-local form
    gu = ops.CSRGraph(dst, src, n_dst, n_src, check_range=False)
    assert gu.regular is None and gu.regular_t is None
    gu.spmm(torch.randn(n_src, 16, device=dev))          # 2.5 MB table: nothing to decide, nothing read back
    assert gu.regular is None and gu._sliced is None
    assert torch.equal(gu.spmm(X), y) and gu.regular is True and gu._sliced is not None
    assert torch.equal(gu.spmm_t(W), dx) and gu.regular_t is True and gu._sliced_t is not None
    su = ops.CSRGraph(skew._S.dst, skew._S.src, n_dst, n_src, check_range=False)
    assert torch.equal(su.spmm(X), ys_plain) and su.regular is False and su._sliced is None and su._S.split is None  # planned
    # (r4) a regular graph of FEW, LONG rows over a table whose slices exceed an L2 (a config-5 edge-scaled shard): virtual
    # rows, so that the launcher's column passes apply — same product
    nl, ns_l, El = 3000, 70_000, 3_000_000  # degree 1000, slices of 8750 rows x 512 B = 4.5 MB
    dl = torch.randint(0, nl, (El,), generator=gen, device=dev, dtype=torch.int32)
    sl_ = torch.randint(0, ns_l, (El,), generator=gen, device=dev, dtype=torch.int32)
    gl = ops.CSRGraph(dl, sl_, nl, ns_l)
    Xl = torch.randn(ns_l, 128, device=dev)
    yl = gl.spmm(Xl)
    assert gl.regular and gl._few_long_rows(128, nl, ns_l) and gl._sliced is None and gl._S.split is not None and not gl._S.split.has_light
    yl_plain = ops.spmm_csr_raw(gl.indptr, gl.indices, None, Xl, plan=gl.plan)
    assert float((yl - yl_plain).abs().max()) <= 1e-5 * float(yl_plain.abs().max())
    assert not gl._few_long_rows(64, nl, ns_l)  # 2.2 MB slices fit an L2: the plain sliced form
    # the decoder's edge-id CSRs never take it: every gathered row is used exactly once
    pairs = ops.EdgePairs(src[:600_000], dst[:600_000], n_src, n_dst)
    pairs.by_src().spmm(torch.randn(600_000, 128, device=dev))
    assert pairs.by_src().regular is False and pairs.by_src()._sliced is None


def test_ops_capture_into_a_hip_graph(dev):
    """The launch path makes no allocation outside torch's allocator and never synchronises, so
    a forward + backward through the ops captures into a HIP graph and replays bit-identically."""
    from dream_gnn_amd import ops

    gen = torch.Generator(device=dev).manual_seed(4)
    n_dst, n_src, E, F = 300, 200, 20000, 128
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
    g = ops.CSRGraph(dst, src, n_dst, n_src, vals=torch.rand(E, generator=gen, device=dev))
    pairs = ops.EdgePairs(src, dst, n_src, n_dst)
    g.transposed(), pairs.by_src(), pairs.by_dst()  # graph construction stays outside the capture
    x = torch.randn(n_src, F, device=dev, requires_grad=True)
    w = torch.randn(n_dst, F, device=dev)

    def work():
        y = ops.spmm_csr(g, x)
        z = ops.gather_concat(pairs, x, y)
        (gx,) = torch.autograd.grad((z * z).sum() + (y * w).sum(), x)
        return y, z, gx

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ref = [t.clone() for t in work()]
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = work()
    for t in outs:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("F", [128, 64, 12, 7])
def test_gather_add_forward_backward(oracle, dev, F):
    """(f2) fused decoder lin1: out[e] = A[src[e]] + B[dst[e]] + bias — one fp32 add chain per
    element, compared exactly against numpy; backward = copy_e -> sum on the SpMM kernel."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(F)
    n_a, n_b, E = 53, 41, 4000
    src = rng.integers(0, n_a, E, dtype=np.int32)
    dst = rng.integers(0, n_b, E, dtype=np.int32)
    A = rng.standard_normal((n_a, F)).astype(np.float32)
    B = rng.standard_normal((n_b, F)).astype(np.float32)
    bias = rng.standard_normal(F).astype(np.float32)
    dO = rng.standard_normal((E, F)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs = ops.EdgePairs(t(src), t(dst), n_a, n_b)
    a, b, c = t(A).requires_grad_(True), t(B).requires_grad_(True), t(bias).requires_grad_(True)
    out = ops.gather_add(pairs, a, b, c)
    out.backward(t(dO))
    assert np.array_equal(out.detach().cpu().numpy(), (A[src] + B[dst]) + bias)  # same fp32 operation order
    assert np.array_equal(ops.gather_add_raw(pairs.src, pairs.dst, t(A), t(B)).cpu().numpy(), A[src] + B[dst])
    for grad, key, n in ((a.grad, src, n_a), (b.grad, dst, n_b)):
        ip, ix, _ = oracle.csr_from_coo(key, np.arange(E, dtype=np.int32), n)
        ref = oracle.spmm_csr(ip, ix, None, dO, acc="f64")
        assert np.abs(grad.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(c.grad.cpu().numpy() - dO.astype(np.float64).sum(0)).max() <= 1e-4 * np.abs(dO).sum(0).max()


@pytest.mark.parametrize("E,keep", [(1, 1), (7, 3), (1000, 1), (1000, 999), (1000, 1000), (1000, 0), (123457, 111111),
                                    (1048575, 524288), (1048576, 943718), (1048577, 1), (1048577, 1048576),
                                    (3_000_000, 2_700_000), (3_000_000, 3_000_000), (3_000_000, 0), (10_000_000, 9_000_000)])
def test_random_subset_mask_bit_exact(oracle, dev, E, keep):
    """(D3) exact-size random edge selection: integer work, bit-exact against the oracle.  Lists below
    2^20 edges are selected by one workgroup (radix select), longer ones by the uniform-window passes
    (keep = 1 / E - 1 at the window's edge of the hash space; keep = E and 0: the direct paths)."""
    from dream_gnn_amd import ops

    for seed in (0, 12345678901234567):
        m = ops.random_subset_mask(E, keep, seed, dev).cpu().numpy()
        assert m.sum() == keep
        assert np.array_equal(m, oracle.random_subset_mask(E, keep, seed))


def test_edge_dropout_select_path_no_sync_and_same_product(dev):
    from dream_gnn_amd import graph as G, ops

    gen = torch.Generator().manual_seed(3)
    n_d, n_s, E = 300, 200, 20000
    d = torch.randint(0, n_d, (E,), generator=gen).to(dev)
    s_ = torch.randint(0, n_s, (E,), generator=gen).to(dev)
    hg = G.HeteroGraph({("drug", "0", "disease"): (d, s_), ("disease", "rev-0", "drug"): (s_, d)},
                       {"drug": n_d, "disease": n_s}).int()
    hg["0"].csr, hg["rev-0"].csr  # parents validated once
    child = G.random_edge_dropout(hg, 0.1, generator=gen)
    rel = child["0"]
    assert isinstance(rel, G.DroppedRelation) and rel._keep_idx is None and rel._mask is None  # select path: an 8-word description, nothing of size E written or read back
    assert rel.desc is not None and tuple(rel.desc.shape) == (8,)
    assert rel.number_of_edges() == int(E * 0.9) and float(rel.keep_mask().sum()) == int(E * 0.9)
    X = torch.randn(n_d, 64, device=dev)
    rebuilt = ops.CSRGraph(rel.dst, rel.src, rel.n_dst, rel.n_src)  # materialises the kept lists (ascending)
    assert rel.src.shape[0] == int(E * 0.9)
    a, b = rel.csr.spmm(X), rebuilt.spmm(X)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    # same generator state -> same subset; the literal randperm procedure is still available
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    m1 = G.random_edge_dropout(hg, 0.3, generator=g1)["0"].keep_mask()
    m2 = G.random_edge_dropout(hg, 0.3, generator=g2)["0"].keep_mask()
    assert torch.equal(m1, m2)
    lit = G.random_edge_dropout(hg, 0.3, selection="randperm")["0"]
    assert lit._keep_idx is not None and lit.number_of_edges() == int(E * 0.7)


@pytest.mark.parametrize("E,keep,off", [(1, 1, 0), (7, 3, 5), (1000, 1, 0), (1000, 999, 17), (1000, 1000, 0), (1000, 0, 3),
                                        (123457, 111111, 1_000_000), (2_000_003, 1_800_002, 0), (5_000_000, 2_500_000, 77),
                                        (5_000_000, 4_999_999, 0), (5_000_000, 1, 0)])
def test_random_subset_description_bit_exact(oracle, dev, E, keep, off):
    """(D3) the 8-word subset description and the mask derived from it: integer work, bit-exact."""
    from dream_gnn_amd import ops

    for seed in (0, 12345678901234567, 2 ** 64 - 1):
        d = ops.random_subset_select(E, keep, seed, dev, e_offset=off)
        assert np.array_equal(d.cpu().numpy(), oracle.random_subset_select(E, keep, seed, off))
        m = ops.keep_mask(d, off + E + 3).cpu().numpy()
        assert m[:off].all() and m[off + E:].all() and m[off:off + E].sum() == keep  # outside the description: kept
        assert np.array_equal(m[off:off + E], oracle.random_subset_mask(E, keep, seed))
    # two descriptions over one edge space (a relation-fused layout)
    a = ops.random_subset_select(E, keep, 5, dev, e_offset=0)
    b = ops.random_subset_select(E, keep, 6, dev, e_offset=E)
    both = ops.keep_mask(torch.stack([a, b]), 2 * E).cpu().numpy()
    assert np.array_equal(both, np.concatenate([oracle.random_subset_mask(E, keep, 5), oracle.random_subset_mask(E, keep, 6)]))
    # two descriptions over the SAME edges (a dropout of a dropped view): an edge survives only if both keep it
    c = ops.random_subset_select(E, keep, 9, dev, e_offset=0)
    nested = ops.keep_mask(torch.stack([a, c]), E).cpu().numpy()
    assert np.array_equal(nested, oracle.random_subset_mask(E, keep, 5) * oracle.random_subset_mask(E, keep, 9))
    assert np.array_equal(nested, oracle.keep_mask(torch.stack([a, c]).cpu().numpy(), E))


@pytest.mark.parametrize("F", [128, 64, 341, 4])
@pytest.mark.parametrize("weighted", [False, True])
def test_edge_dropout_on_the_fly_every_kernel(oracle, dev, F, weighted):
    """`keep(eid[p])` evaluated inside the kernels == the product over a graph REBUILT from the kept
    edges (the reference's construction, augmentation.py:48-65), for every kernel form and the
    transpose; dropped edges are skipped, so Inf in a source row that only dropped edges touch
    does not reach the output (a 0/1 value mask would give 0 * Inf = NaN)."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(F + weighted)
    n_dst, n_src, E = 300, 2500, 6000  # many sources of degree 1-3
    dst = rng.integers(0, n_dst, E).astype(np.int32)
    src = rng.integers(0, n_src, E).astype(np.int32)
    vals = rng.standard_normal(E).astype(np.float32) if weighted else None
    keep_n = max(1, int(E * 0.7))
    mask = oracle.random_subset_mask(E, keep_n, 77).astype(bool)
    kept_per_src = np.bincount(src[mask], minlength=n_src)
    dead = np.flatnonzero((np.bincount(src, minlength=n_src) > 0) & (kept_per_src == 0))
    assert dead.size > 50
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    X_inf = X.copy()
    X_inf[dead[:25]] = np.inf
    X_inf[dead[25:50]] = np.nan
    ss = rng.uniform(0.5, 1.5, n_src).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, n_dst).astype(np.float32)
    ip, ix, e0 = oracle.csr_from_coo(dst[mask], src[mask], n_dst)
    v0 = None if vals is None else vals[mask][e0]
    ref = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="f64")
    bound = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="abs")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(dst), t(src), n_dst, n_src, vals=t(vals))
    desc = ops.random_subset_select(E, keep_n, 77, dev)
    view = g.dropped(desc)
    assert view.indptr is g.indptr and torch.equal(view.keep_mask().cpu(), torch.from_numpy(mask.astype(np.float32)))
    Xd, ssd, dsd = t(X_inf), t(ss), t(ds)
    forms = {"wave-per-row": ops.spmm_csr_raw(g.indptr, g.indices, g.vals, Xd, ssd, dsd, eid=g.eid, keep=desc),
             "planned": ops.spmm_csr_raw(g.indptr, g.indices, g.vals, Xd, ssd, dsd, plan=g.plan, eid=g.eid, keep=desc),
             "CSRGraph view": view.spmm(Xd, ssd, dsd)}
    if F % 4 == 0:
        sl = ops.SlicedCSR(t(dst), t(src), n_dst, n_src, vals=t(vals))
        forms["xcd-sliced"] = sl.spmm(Xd, ssd, dsd, keep=desc)
    for name, y in forms.items():
        y = y.cpu().numpy()
        assert np.isfinite(y).all(), name
        assert np.all(np.abs(y - ref) <= RTOL * bound + 1e-30), name
    # transpose: dX = diag(ss) A_kept^T diag(ds) W
    W = rng.standard_normal((n_dst, F)).astype(np.float32)
    tp, ti, te = oracle.csr_from_coo(src[mask], dst[mask], n_src)
    vt = None if vals is None else vals[mask][te]
    ref_t = oracle.spmm_csr(tp, ti, vt, W, ds, ss, acc="f64")
    bound_t = oracle.spmm_csr(tp, ti, vt, W, ds, ss, acc="abs")
    dx = view.spmm_t(t(W), ssd, dsd).cpu().numpy()
    assert np.all(np.abs(dx - ref_t) <= RTOL * bound_t + 1e-30)
    assert np.all(dx[dead] == 0)
    # autograd through the view
    x = t(X).requires_grad_(True)
    ops.spmm_csr(view, x, ssd, dsd).backward(t(W))
    assert np.all(np.abs(x.grad.cpu().numpy() - ref_t) <= RTOL * bound_t + 1e-30)


def test_subset_selection_window_miss_is_taken_over_exactly(oracle, dev):
    """Long lists find the threshold inside an 8-sigma window of the hash space; if the window ever misses,
    one workgroup selects the list on its own.  With the window narrowed to ~2 edges it (almost) always
    misses: the description is still the oracle's, bit for bit."""
    from dream_gnn_amd import ops

    from dream_gnn_amd import _lib

    _lib.set_tuning("select_narrow_window", 1)  # (the library reads no environment on a launch path)
    try:
        for E, keep, seed in ((1_500_000, 1_350_000, 3), (2_000_000, 1_000_000, 2 ** 63 + 5), (1_200_000, 7, 11)):
            d = ops.random_subset_select(E, keep, seed, dev)
            assert np.array_equal(d.cpu().numpy(), oracle.random_subset_select(E, keep, seed, 0))
    finally:
        _lib.set_tuning("select_narrow_window", 0)


def test_batched_subset_selection_equals_single_calls(oracle, dev):
    """8 subsets from one series of launches (dgmi_random_subset_select_batch) == 8 single selections
    == the oracle, with different sizes, offsets and an empty keep among them."""
    from dream_gnn_amd import ops

    Es = [1, 7, 1000, 465_000, 2_746, 4_100_003, 50, 2_000_000, 33, 12]  # 10 lists: two batches, short and long lists mixed
    keeps = [1, 3, 900, 418_500, 2_471, 0, 50, 1_800_000, 1, 11]
    seeds = [0, 1, 2 ** 64 - 1, 12345678901234567, 5, 6, 7, 2 ** 63, 9, 10]
    offs = [0, 5, 0, 1000, 0, 7, 0, 0, 3, 0]
    d = ops.random_subset_select_batch(Es, keeps, seeds, dev, offs)
    assert tuple(d.shape) == (10, 8)
    for i in range(10):
        want = oracle.random_subset_select(Es[i], keeps[i], seeds[i], offs[i])
        assert np.array_equal(d[i].cpu().numpy(), want), i
        assert torch.equal(d[i], ops.random_subset_select(Es[i], keeps[i], seeds[i], dev, e_offset=offs[i])), i


@pytest.mark.parametrize("F", [128, 341, 64])
@pytest.mark.parametrize("act,slope", [(1, 0.1), (1, 0.0), (0, 0.0)])
def test_output_epilogue_in_every_kernel_form(dev, F, act, slope):
    """f3 epilogue (`dropout(agg_act(.))`, layers.py:134-138) inside the kernel that writes Y: the
    fused result equals activation + mask applied afterwards to the un-fused product, bit for bit,
    for the wave-per-row, planned (incl. chunked long rows) and XCD-sliced forms; the backward
    equals torch autograd of the un-fused composition."""
    from dream_gnn_amd import ops

    gen = torch.Generator().manual_seed(F + act)
    n_dst, n_src, E = 150, 220, 9000
    dst = torch.randint(0, n_dst, (E,), generator=gen, dtype=torch.int32)
    dst[:1500] = 7  # a row long enough to be cut into chunks by the plan
    dst[dst == 3] = 4  # an empty row
    src = torch.randint(0, n_src, (E,), generator=gen, dtype=torch.int32)
    g = ops.CSRGraph(dst.to(dev), src.to(dev), n_dst, n_src)
    X = torch.randn(n_src, F, generator=gen).to(dev)
    ss, ds = torch.rand(n_src, generator=gen).to(dev) + 0.5, torch.rand(n_dst, generator=gen).to(dev) + 0.5
    mask = (torch.rand(n_dst, F, generator=gen) < 0.7).float().to(dev)
    mscale = 1.0 / 0.7
    epi = (act, slope, mask, mscale)

    def post(y):
        y = torch.nn.functional.leaky_relu(y, slope) if act == 1 else y
        return y * mask * mscale

    forms = {"wave-per-row": (ops.spmm_csr_raw(g.indptr, g.indices, None, X, ss, ds),
                              ops._launch_spmm(dev, g.indptr, g.indices, None, X, ss, ds, None, None, n_dst, n_src, F, F, epi=epi)),
             "planned": (ops.spmm_csr_raw(g.indptr, g.indices, None, X, ss, ds, plan=g.plan),
                         ops._launch_spmm(dev, g.indptr, g.indices, None, X, ss, ds, None, g.plan, n_dst, n_src, F, F, epi=epi))}
    if F % 4 == 0:
        sl = ops.SlicedCSR(dst.to(dev), src.to(dev), n_dst, n_src)
        forms["xcd-sliced"] = (sl.spmm(X, ss, ds), sl.spmm(X, ss, ds, epi=epi))
    for name, (plain, fused) in forms.items():
        assert torch.equal(fused, post(plain)), name
    # autograd
    x1 = X.clone().requires_grad_(True)
    y1 = ops.spmm_csr_act_dropout(g, x1, ss, ds, act, slope, mask, mscale)
    x2 = X.clone().requires_grad_(True)
    y2 = post(ops.spmm_csr(g, x2, ss, ds))
    W = torch.randn(n_dst, F, generator=gen).to(dev)
    y1.backward(W)
    y2.backward(W)
    assert torch.equal(y1, y2)
    assert float((x1.grad - x2.grad).abs().max()) <= 1e-6 * float(x2.grad.abs().max())


@pytest.mark.parametrize("N,D,k", [(763, 768, 4), (681, 768, 4), (33, 8, 1), (1000, 64, 16), (5000, 768, 8), (32, 1024, 3),
                                   (97, 24, 13), (1536, 64, 16), (2048, 64, 1), (3001, 40, 2), (16500, 16, 5), (20000, 768, 4), (9001, 200, 16), (50001, 72, 6), (41000, 64, 12), (25000, 136, 3),
                                   (3000, 64, 64), (30000, 128, 33), (60000, 72, 40),
                                   (50000, 40, 3), (52000, 136, 5), (49152, 768, 4), (70001, 200, 16), (50200, 200, 64)])
def test_fused_knn_kernel_finds_the_k_most_similar_rows(dev, N, D, k):
    """(f4) `dgmi_knn_cosine_topk_f32` (fp32 MFMA tiles + running top-k on chip) against a brute-force
    float64 similarity matrix: every row's selected neighbours are k distinct valid ids whose
    similarities are the k largest (up to fp32 rounding of near-ties), in descending order, self first."""
    from dream_gnn_amd import graph as G, ops

    gen = torch.Generator().manual_seed(N + D + k)
    X = torch.randn(N, D, generator=gen)
    X[3] = X[2] * 2.0  # an exact duplicate direction: similarity 1 with another row
    xn = (X / X.norm(dim=1, keepdim=True)).to(dev)
    assert ops.knn_cosine_supported(N, D, k)
    # N < 1536: fp32 kernel, candidates split over workgroups + merge; above: bf16 screen + exact rescoring
    # (full rectangle of tile pairs up to ~24k rows / ~41k at k > 8, triangular sweep above; 256 x 256 tiles from 49152:
    # the phase-interleaved LDS-DMA kernel with 2+ K chunks (D = 72, 136, 200, 768 here: 2, 3, 4, 12 chunks; k up to 64),
    # the register-staged one for one-chunk rows (D = 40))
    nbr = ops.knn_cosine_topk(xn, k).long()
    assert nbr.shape == (N, k) and int(nbr.min()) >= 0 and int(nbr.max()) < N
    # N >= 49152 (256 x 256 screen tiles): check the first / last / a random 1500 rows against the brute force
    rows = torch.arange(N, device=dev)
    if N > 25000:
        rows = torch.cat([rows[:1500], rows[-1500:], torch.randperm(N, generator=gen)[:1500].to(dev)])
    sim = xn[rows].double() @ xn.double().t()
    sel = nbr[rows]
    assert all(len(set(r.tolist())) == k for r in sel)  # distinct
    got = torch.gather(sim, 1, sel)
    want = torch.topk(sim, k, dim=1).values
    assert float((got - want).abs().max()) <= 2e-6  # same similarity multiset (near-ties may swap ids)
    assert bool((got[:, :-1] >= got[:, 1:] - 2e-6).all()) if k > 1 else True  # descending
    self_or_dup = (sel[:, 0] == rows) | ((got[:, 0] - 1.0).abs() < 1e-6)
    assert bool(self_or_dup.all())
    # the graph builder uses it and agrees with the torch GEMM + top-k path on the similarities it keeps
    a = G.feature_similarity_graph(X.to(dev), k, fused=True).coalesce()
    b = G.feature_similarity_graph(X.to(dev), k, fused=False).coalesce()
    assert a.shape == b.shape
    if torch.equal(a.indices(), b.indices()):
        assert torch.equal(a.values(), b.values())
    else:  # near-ties picked differently: the graphs still have the same size up to those rows
        assert abs(a._nnz() - b._nnz()) <= 4


def test_screened_knn_recomputes_overflowing_rows_exactly(dev):
    """(f4) the bf16 screen keeps at most 64 candidates per (query, candidate split); a cluster of 400
    near-identical rows puts far more than that within the 2-eps margin of the k-th best, so those queries
    are flagged and their tiles recomputed by the fp32 kernel — the answer is still the exact top-k."""
    from dream_gnn_amd import ops

    # full rectangle / triangular sweep (both directions overflow) / k > 16: the exact-row take-over kernel
    # ... and the 256 x 256 LDS-DMA kernel (N >= 49152, D > 64): its wave lists fill (the groups of a cluster in one
    # quadrant), the regions of the clustered queries overflow in both directions, everything else goes through the pool
    for N, D, k in ((10000, 128, 4), (30000, 64, 4), (9000, 64, 40), (50000, 96, 4), (51000, 200, 12)):
        gen = torch.Generator().manual_seed(5)
        X = torch.randn(N, D, generator=gen)
        X[3000:3400] = X[3000] + 1e-3 * torch.randn(400, D, generator=gen)
        if k > 16:  # enough near-duplicates to overflow the wider regions too
            X[1000:2600] = X[1000] + 1e-3 * torch.randn(1600, D, generator=gen)
        X[N - 700:N - 100] = X[N - 1] + 1e-3 * torch.randn(600, D, generator=gen)
        xn = (X / X.norm(dim=1, keepdim=True)).to(dev)
        nbr = ops.knn_cosine_topk(xn, k).long()
        worst = 0.0
        for lo in range(0, N, 10000):  # (row blocks: the float64 matrix of 50 000 rows would be 20 GB)
            sim = xn[lo:lo + 10000].double() @ xn.double().t()
            got = torch.gather(sim, 1, nbr[lo:lo + 10000])
            want = torch.topk(sim, k, dim=1).values
            worst = max(worst, float((got - want).abs().max()))
            del sim
        assert worst <= 2e-6
        assert all(len(set(r.tolist())) == k for r in nbr[2990:3410].cpu())


def test_adjacency_assembly_without_library_sort_or_atomics_is_bit_identical(dev):
    """(r4, D2) `graph._normalized_adjacency` at scale: entries coalesced through the library's own record sort and row sums as
    differences of a running sum — against the literal torch pipeline (`coalesce()` + float64 `index_add_`, data_loader.py:297-308,
    utils.py:11-27): the same indices and the same values, bit for bit (duplicates, self loops, an isolated node)."""
    from dream_gnn_amd import graph as G

    gen = torch.Generator().manual_seed(11)
    for n, k in ((40000, 9), (3000, 100)):
        nbr = torch.randint(0, n - 1, (n, k), generator=gen)  # (node n - 1 is nobody's neighbour)
        nbr[:, 1] = nbr[:, 0]                                  # duplicates inside a row
        nbr[5] = 5                                             # self loops
        rows = torch.arange(n).repeat_interleave(k).to(dev)
        cols = nbr.reshape(-1).to(dev)
        assert 2 * rows.numel() + n >= 2 ** 18
        got = G._normalized_adjacency(rows, cols, n, True)
        eye = torch.arange(n, device=dev)
        r, c = torch.cat([rows, cols, eye]), torch.cat([cols, rows, eye])
        adj = torch.sparse_coo_tensor(torch.stack([r, c]), torch.ones(r.numel(), dtype=torch.float64, device=dev), (n, n)).coalesce()
        idx, val = adj.indices(), adj.values()
        rowsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, idx[0], val)
        want = (val * (1.0 / rowsum)[idx[0]]).to(torch.float32)
        assert torch.equal(got._indices(), idx) and torch.equal(got._values(), want)


def test_screened_knn_pool_that_runs_out_and_first_kernel_agree(dev):
    """(r4, f4) The 256 x 256 LDS-DMA screen kernel files what it keeps into a pool of record chunks; with the pool cut to a
    sliver (tuning knob) almost every record takes the direct-append fallback instead — the answer is the same exact top-k,
    and the same as the first (register-staged) kernel's."""
    from dream_gnn_amd import _lib, ops

    N, D, k = 50000, 96, 5
    gen = torch.Generator().manual_seed(77)
    X = torch.randn(N, D, generator=gen)
    X[100:140] = X[100] + 1e-3 * torch.randn(40, D, generator=gen)  # a small cluster: no region overflows, lists get busy
    xn = (X / X.norm(dim=1, keepdim=True)).to(dev)
    want = ops.knn_cosine_topk(xn, k).long()
    rows = torch.cat([torch.arange(2000), torch.arange(N - 1000, N)]).to(dev)
    sim = xn[rows].double() @ xn.double().t()
    ref = torch.topk(sim, k, dim=1).values
    assert float((torch.gather(sim, 1, want[rows]) - ref).abs().max()) <= 2e-6
    for knob, value in (("knn_pool_chunks", 48), ("knn_screen_first", 1)):
        _lib.set_tuning(knob, value)
        try:
            got = ops.knn_cosine_topk(xn, k).long()
        finally:
            _lib.set_tuning(knob, 0)
        assert float((torch.gather(sim, 1, got[rows]) - ref).abs().max()) <= 2e-6, knob
        same = (got == want).all(dim=1).float().mean()
        assert float(same) > 0.999, knob  # (near-ties may pick another id)


def test_compaction_invert_and_nested_descriptions_randomized(oracle, dev):
    """(r3) The on-the-fly dropout path after round 3 — survivors compacted to the front of every 64-id batch, INVERTED
    descriptions (the complement form subtracts a relation's dropped edges), several descriptions over the same edges
    ANDed — on 80 random small shapes: odd widths (dword kernel), rows shorter / longer than a batch, keep = 0 and
    keep = E, empty rows, planned and wave-per-row launches, against the oracle on exactly the surviving edges."""
    from dream_gnn_amd import ops

    rng = np.random.default_rng(2026)
    for case in range(80):
        n_dst, n_src = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        E = int(rng.integers(1, 900))
        F = int(rng.choice([1, 3, 4, 7, 8, 12, 33, 64, 65, 128]))
        dst = rng.integers(0, n_dst, E).astype(np.int32)
        if case % 5 == 0:
            dst[: E // 2] = 0  # one row far longer than a 64-id batch
        src = rng.integers(0, n_src, E).astype(np.int32)
        weighted = bool(case % 2)
        vals = rng.standard_normal(E).astype(np.float32) if weighted else None
        X = rng.standard_normal((n_src, F)).astype(np.float32)
        ss = rng.uniform(0.5, 1.5, n_src).astype(np.float32) if case % 3 else None
        ds = rng.uniform(0.5, 1.5, n_dst).astype(np.float32) if case % 4 else None
        keep1 = int(rng.choice([0, 1, E // 3, E - 1, E])) if case % 7 == 0 else int(rng.integers(0, E + 1))
        d1 = oracle.random_subset_select(E, keep1, 100 + case, 0).copy()
        descs = [d1]
        if case % 3 == 0:  # inverted: the edges the description drops take part
            d1[6] = 1
        if case % 4 == 1:  # a second description over the same edges: intersection
            descs.append(oracle.random_subset_select(E, int(rng.integers(0, E + 1)), 500 + case, 0).copy())
        table = np.stack(descs)
        mask = oracle.keep_mask(table, E).astype(bool)
        t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
        g = ops.CSRGraph(t(dst), t(src), n_dst, n_src, vals=t(vals))
        keep_t = t(table)
        assert np.array_equal(ops.keep_mask(keep_t, E).cpu().numpy().astype(bool), mask), case
        ip, ix, e0 = oracle.csr_from_coo(dst[mask], src[mask], n_dst)
        v0 = None if vals is None else vals[mask][e0]
        ref = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="f64")
        bound = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="abs")
        for plan in (None, g.plan):
            y = ops.spmm_csr_raw(g.indptr, g.indices, g.vals, t(X), t(ss), t(ds), plan=plan, eid=g.eid, keep=keep_t)
            err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
            assert np.all(err <= RTOL * bound + 1e-30), (case, "planned" if plan is not None else "wave-per-row", F, E, keep1)


@pytest.mark.parametrize("shape", [(300, 2500, 6000), (1, 5, 1), (40, 30, 0), (7, 9, 2047), (7, 9, 2048), (7, 9, 2049),
                                   (2000, 100, 70000), (50, 4000, 131072), (3, 3, 64), (9000, 500, 40000)])
@pytest.mark.parametrize("weighted", [False, True])
def test_compacted_layout_is_the_layout_of_the_kept_edge_list(oracle, dev, shape, weighted):
    """(r4) `dgmi_compact_layout_i32`: the per-step replacement for the reference's graph rebuild (augmentation.py:48-65)
    at scale.  CSR and XCD-sliced layouts (8 and 3 slices) of the parent, compacted under plain, inverted and nested
    descriptions, are BIT-IDENTICAL to the same layouts built from the kept edge list — pointers, ids and values
    (tile edges 2047 / 2048 / 2049, keep = 0, keep = E, empty rows, an empty graph)."""
    from dream_gnn_amd import _lib, ops

    n_dst, n_src, E = shape
    rng = np.random.default_rng(E + weighted)
    dst = rng.integers(0, n_dst, E).astype(np.int32)
    src = rng.integers(0, n_src, E).astype(np.int32)
    if E > 100:
        dst[dst == 1] = 0  # an empty row
    vals = rng.standard_normal(E).astype(np.float32) if weighted else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    tables = []
    for keep_n, inv in ((int(E * 0.9), 0), (0, 0), (E, 0), (E // 3, 1)):
        d = oracle.random_subset_select(E, keep_n, 40 + keep_n, 0).copy()
        d[6] = inv
        tables.append(np.stack([d]))
    tables.append(np.stack([oracle.random_subset_select(E, int(E * 0.8), 1, 0), oracle.random_subset_select(E, int(E * 0.5), 2, 0)]))
    if E > 10:  # a description that covers only a window of the edge ids (relation-fused layouts)
        tables.append(np.stack([oracle.random_subset_select(E // 2, E // 4, 3, E // 4)]))
    for table in tables:
        mask = oracle.keep_mask(table, E).astype(bool)
        nk = int(mask.sum())
        keep_t = t(table)
        # plain CSR
        ip, ix, ei = oracle.csr_from_coo(dst, src, n_dst)
        v_csr = None if vals is None else vals[ei]
        want_ptr, want_ix, want_v = oracle.compact_layout(ip, ix, v_csr, ei, table)
        kp, ki, _ = oracle.csr_from_coo(dst[mask], src[mask], n_dst)  # the reference's construction
        assert np.array_equal(want_ptr, kp) and np.array_equal(want_ix, ki)
        ptr_o, idx_o, v_o = _lib.torch_ops.compact_layout(t(ip), t(ix), t(v_csr), t(ei), keep_t)
        assert np.array_equal(ptr_o.cpu().numpy(), kp)
        assert np.array_equal(idx_o.cpu().numpy()[:nk], ki)
        if weighted:
            assert np.array_equal(v_o.cpu().numpy()[:nk], want_v)
        # XCD-sliced
        for n_slices in (8, 3):
            sl = ops.SlicedCSR(t(dst), t(src), n_dst, n_src, vals=t(vals), n_slices=n_slices)
            c = sl.compacted(keep_t, sl.vals)
            sp, si, se = oracle.csr_sliced_from_coo(dst[mask], src[mask], n_dst, n_src, n_slices)
            assert np.array_equal(c.segptr.cpu().numpy(), sp), (n_slices, nk)
            assert np.array_equal(c.indices.cpu().numpy()[:nk], si)
            if weighted:
                assert np.array_equal(c.vals.cpu().numpy()[:nk], vals[mask][se])


@pytest.mark.parametrize("F", [128, 64, 256])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("compact", [True, False])
def test_edge_dropped_view_of_a_sliced_graph_compacted_or_on_the_fly(oracle, dev, monkeypatch, F, weighted, compact):
    """A dropped view of a graph that takes the XCD-local form: with compaction (default) and with the on-the-fly
    KEEP kernels (DGMI_COMPACT_DROPPED=0) the product, its transpose and autograd equal the product over a CSR rebuilt
    from the kept edges; Inf / NaN in source rows that only dropped edges touch never reach the output; the
    compacted layouts are made once per view and layout."""
    from dream_gnn_amd import ops

    monkeypatch.setattr(ops, "FORCE_KERNEL", "sliced")
    monkeypatch.setattr(ops, "COMPACT_DROPPED", compact)
    rng = np.random.default_rng(F + 2 * weighted)
    n_dst, n_src, E = 700, 9000, 30000
    dst = rng.integers(0, n_dst, E).astype(np.int32)
    src = rng.integers(0, n_src, E).astype(np.int32)
    vals = rng.standard_normal(E).astype(np.float32) if weighted else None
    keep_n = max(1, int(E * 0.7))
    mask = oracle.random_subset_mask(E, keep_n, 77).astype(bool)
    dead = np.flatnonzero((np.bincount(src, minlength=n_src) > 0) & (np.bincount(src[mask], minlength=n_src) == 0))
    assert dead.size > 50
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    X_bad = X.copy()
    X_bad[dead[:25]] = np.inf
    X_bad[dead[25:50]] = np.nan
    ss = rng.uniform(0.5, 1.5, n_src).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, n_dst).astype(np.float32)
    ip, ix, e0 = oracle.csr_from_coo(dst[mask], src[mask], n_dst)
    v0 = None if vals is None else vals[mask][e0]
    ref = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="f64")
    bound = oracle.spmm_csr(ip, ix, v0, X, ss, ds, acc="abs")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(dst), t(src), n_dst, n_src, vals=t(vals))
    view = g.dropped(ops.random_subset_select(E, keep_n, 77, dev))
    y = view.spmm(t(X_bad), t(ss), t(ds)).cpu().numpy()
    assert np.isfinite(y).all() and np.all(np.abs(y - ref) <= RTOL * bound + 1e-30)
    assert ("sliced" in view._c) == compact
    first = view._c.get("sliced")
    assert np.array_equal(view.spmm(t(X_bad), t(ss), t(ds)).cpu().numpy(), y) and view._c.get("sliced") is first
    W = rng.standard_normal((n_dst, F)).astype(np.float32)
    tp, ti, te = oracle.csr_from_coo(src[mask], dst[mask], n_src)
    vt = None if vals is None else vals[mask][te]
    ref_t = oracle.spmm_csr(tp, ti, vt, W, ds, ss, acc="f64")
    bound_t = oracle.spmm_csr(tp, ti, vt, W, ds, ss, acc="abs")
    dx = view.spmm_t(t(W), t(ss), t(ds)).cpu().numpy()
    assert np.all(np.abs(dx - ref_t) <= RTOL * bound_t + 1e-30) and np.all(dx[dead] == 0)
    x = t(X).requires_grad_(True)
    ops.spmm_csr(view, x, t(ss), t(ds)).backward(t(W))
    assert np.all(np.abs(x.grad.cpu().numpy() - ref_t) <= RTOL * bound_t + 1e-30)
    # the un-dropped parent is untouched by its views
    ip_f, ix_f, e_f = oracle.csr_from_coo(dst, src, n_dst)
    ref_f = oracle.spmm_csr(ip_f, ix_f, None if vals is None else vals[e_f], X, ss, ds, acc="f64")
    assert np.abs(g.spmm(t(X), t(ss), t(ds)).cpu().numpy() - ref_f).max() <= RTOL * np.abs(ref_f).max()
    # a second dropout on the view (descriptions ANDed): its own compacted layout
    mask2 = mask & oracle.random_subset_mask(E, E // 2, 5).astype(bool)
    view2 = view.dropped(ops.random_subset_select(E, E // 2, 5, dev))
    ip2, ix2, e2 = oracle.csr_from_coo(dst[mask2], src[mask2], n_dst)
    ref2 = oracle.spmm_csr(ip2, ix2, None if vals is None else vals[mask2][e2], X, ss, ds, acc="f64")
    assert np.abs(view2.spmm(t(X), t(ss), t(ds)).cpu().numpy() - ref2).max() <= RTOL * max(np.abs(ref2).max(), 1e-30)


def _reference_adjacency(rng, n, k, dup_self=False):
    """`normalize(adj + adj.T + I)` of a 0/1 kNN matrix as the reference builds it (data_loader.py:297-308,
    utils.py:11-17: float64 row sums, values cast to fp32), neighbours random; COO (row, col, val), row-major."""
    import scipy.sparse as sp

    rows = np.repeat(np.arange(n), k)
    cols = rng.integers(0, n, n * k)
    if dup_self:
        cols[::k] = rows[::k]  # self in the top-k, as a cosine kNN always has it: diagonal multiplicity 3
    a = sp.coo_matrix((np.ones(n * k), (rows, cols)), shape=(n, n)).tocsr()
    a.data[:] = 1.0  # a 0/1 matrix (duplicate picks collapse)
    a = a + a.T + sp.eye(n)
    r_inv = np.power(np.asarray(a.sum(1)).ravel(), -1.0)
    a = sp.diags(r_inv).dot(a).tocoo()
    return a.row.astype(np.int32), a.col.astype(np.int32), a.data.astype(np.float32)


def test_row_multiplicity_finds_the_reference_adjacency_form_and_only_that(dev):
    """`dgmi_row_multiplicity_f32`: an adjacency built the reference's way is row scale x m (m in 1..3 here) to 2 ulp in
    every row; arbitrary values, a zero / negative value or a multiplicity of 9 in ONE row raise the flag."""
    from dream_gnn_amd import _lib, ops

    rng = np.random.default_rng(3)
    n = 3000
    r, c, v = _reference_adjacency(rng, n, 6, dup_self=True)
    t = lambda a: torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(r), t(c), n, n, vals=t(v))
    scale, code, fail = _lib.torch_ops.row_multiplicity(g.indptr, g.vals, ops.MULT_REL_TOL)
    assert int(fail.item()) == 0
    code, scale = code.cpu().numpy(), scale.cpu().numpy()
    assert code.min() == 0 and code.max() == 2 and (code == 2).sum() >= n  # the diagonal: A + A^T + I = 3
    rows = np.repeat(np.arange(n), np.diff(g.indptr.cpu().numpy()))
    vals_csr = g.vals.cpu().numpy()
    assert np.all(np.abs((code + 1) * scale[rows] - vals_csr) <= 2.4e-7 * vals_csr)
    mf = g._mult_form()
    assert mf is not None and mf[0] == "row" and torch.equal(mf[2][g.eid.long()].cpu(), torch.from_numpy(code))
    gt = ops.CSRGraph(t(c), t(r), n, n, vals=t(v))  # the transpose handed in as a graph of its own: M D^-1, a COLUMN scale
    mft = gt._mult_form()
    assert mft is not None and mft[0] == "col" and torch.equal(mft[1], mf[1]) and torch.equal(mft[2], mf[2])
    for name, v2 in (("random", rng.uniform(0.1, 1.0, v.size).astype(np.float32)),
                     ("one zero", np.where(np.arange(v.size) == 17, 0.0, v).astype(np.float32)),
                     ("one negative", np.where(np.arange(v.size) == 17, -v, v).astype(np.float32)),
                     ("one off by 1e-5", np.where(np.arange(v.size) == 17, v * (1 + 1e-5), v).astype(np.float32))):
        g2 = g.with_values(t(v2))
        assert int(_lib.torch_ops.row_multiplicity(g2.indptr, g2.vals, ops.MULT_REL_TOL)[2].item()) == 1, name
        assert g2._mult_form() is None, name
    # multiplicities up to 8 are carried, 9 is not
    r3 = np.concatenate([np.zeros(9, np.int32), np.array([1, 1], np.int32)])
    c3 = np.arange(11, dtype=np.int32)
    for top, want in ((8, 0), (9, 1)):
        v3 = np.concatenate([np.arange(1, 10, dtype=np.float32).clip(max=top) * np.float32(0.01), [0.5, 0.5]]).astype(np.float32)
        v3[8] = top * np.float32(0.01)
        g3 = ops.CSRGraph(t(r3), t(c3), 3, 11, vals=t(v3))  # row 2 is empty
        assert int(_lib.torch_ops.row_multiplicity(g3.indptr, g3.vals, ops.MULT_REL_TOL)[2].item()) == want


@pytest.mark.parametrize("F", [128, 64, 256])
@pytest.mark.parametrize("scaled", [False, True])
def test_weighted_product_without_a_value_stream_equals_the_weighted_product(oracle, dev, monkeypatch, F, scaled):
    """(r4) An adjacency in the reference's format through the XCD-local kernels with the multiplicity in the id words
    and the row scale as a diagonal scale: every row of the product and of its transpose within 1e-5 of the f64 oracle
    ON THE ORIGINAL VALUES (and of on-device `torch.spmm`), un-dropped, edge-dropped after compaction and edge-dropped
    on the fly; the same graph with DGMI_MULT_FORM off (value stream) agrees to fp32 rounding; an arbitrary-valued
    graph keeps its value stream."""
    from dream_gnn_amd import ops

    monkeypatch.setattr(ops, "FORCE_KERNEL", "sliced")
    rng = np.random.default_rng(F + scaled)
    n = 2500
    r, c, v = _reference_adjacency(rng, n, 8, dup_self=True)
    E = r.size
    X = rng.standard_normal((n, F)).astype(np.float32)
    W = rng.standard_normal((n, F)).astype(np.float32)
    ss = rng.uniform(0.5, 1.5, n).astype(np.float32) if scaled else None
    ds = rng.uniform(0.5, 1.5, n).astype(np.float32) if scaled else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    g = ops.CSRGraph(t(r), t(c), n, n, vals=t(v))
    assert g._mult_form() is not None
    # the transposed adjacency as a graph of its own (column-scale form): its product is the first graph's transpose product
    gt = ops.CSRGraph(t(c), t(r), n, n, vals=t(v))
    assert gt._mult_form()[0] == "col"
    assert float((gt.spmm(t(W), t(ds), t(ss)) - g.spmm_t(t(W), t(ss), t(ds))).abs().max()) <= 2e-6 * float(g.spmm_t(t(W), t(ss), t(ds)).abs().max())
    assert float((gt.spmm_t(t(X), t(ds), t(ss)) - g.spmm(t(X), t(ss), t(ds))).abs().max()) <= 2e-6 * float(g.spmm(t(X), t(ss), t(ds)).abs().max())
    assert "sliced" not in gt._v and "sliced_t" not in gt._v
    keep_n = int(E * 0.9)
    desc = ops.random_subset_select(E, keep_n, 7, dev)
    full = np.ones(E, bool)
    dropped = oracle.random_subset_mask(E, keep_n, 7).astype(bool)

    def refs(mask):
        ip, ix, e0 = oracle.csr_from_coo(r[mask], c[mask], n)
        y = oracle.spmm_csr(ip, ix, v[mask][e0], X, ss, ds, acc="f64"), oracle.spmm_csr(ip, ix, v[mask][e0], X, ss, ds, acc="abs")
        tp, ti, te = oracle.csr_from_coo(c[mask], r[mask], n)
        dx = oracle.spmm_csr(tp, ti, v[mask][te], W, ds, ss, acc="f64"), oracle.spmm_csr(tp, ti, v[mask][te], W, ds, ss, acc="abs")
        return y, dx

    for mask, view, compact in ((full, g, True), (dropped, g.dropped(desc), True), (dropped, g.dropped(desc), False)):
        monkeypatch.setattr(ops, "COMPACT_DROPPED", compact)
        (y64, yabs), (dx64, dxabs) = refs(mask)
        y = view.spmm(t(X), t(ss), t(ds)).cpu().numpy()
        dx = view.spmm_t(t(W), t(ss), t(ds)).cpu().numpy()
        assert np.all(np.abs(y - y64) <= RTOL * yabs + 1e-30) and np.all(np.abs(dx - dx64) <= RTOL * dxabs + 1e-30)
        if view is not g and compact:
            assert view._c["sliced"].id_mult and view._c["sliced"].vals is None
    assert "sliced" not in g._v and "sliced/ids" in g._v  # the value stream of this layout was never built
    if not scaled:  # the reference's own call, on the device
        adj = torch.sparse_coo_tensor(torch.stack([t(r).long(), t(c).long()]), t(v), (n, n))
        y_t = torch.spmm(adj, t(X))
        assert float((g.spmm(t(X)) - y_t).abs().max()) <= RTOL * float(y_t.abs().max())
    # value-stream form of the same graph
    monkeypatch.setattr(ops, "MULT_FORM", False)
    gv = ops.CSRGraph(t(r), t(c), n, n, vals=t(v))
    yv = gv.spmm(t(X), t(ss), t(ds))
    assert gv._mult_form() is None and "sliced" in gv._v
    assert float((yv - g.spmm(t(X), t(ss), t(ds))).abs().max()) <= RTOL * float(yv.abs().max())
    monkeypatch.setattr(ops, "MULT_FORM", True)
    # arbitrary values: value stream, same answer as the oracle
    va = rng.standard_normal(E).astype(np.float32)
    ga = ops.CSRGraph(t(r), t(c), n, n, vals=t(va))
    ip, ix, e0 = oracle.csr_from_coo(r, c, n)
    ya = ga.spmm(t(X)).cpu().numpy()
    assert ga._mult_form() is None and "sliced" in ga._v
    assert np.all(np.abs(ya - oracle.spmm_csr(ip, ix, va[e0], X, acc="f64")) <= RTOL * oracle.spmm_csr(ip, ix, va[e0], X, acc="abs") + 1e-30)
