"""The torch operator layer of SURVEY §8(b): `torch.ops.dreamgnn_mi.*` (csrc/dgmi_torch.cpp) — dispatcher
ops over the C ABI, with autograd registered for `spmm_csr`.  Same parity cases as the C-ABI tests:
golden vectors produced by the reference's own call sites (th.spmm by the installed ATen; copy_u by
ATen on unit values and by the DGL stand-in)."""
import numpy as np
import pytest
import torch

import _cases as C

pytestmark = pytest.mark.gpu
OPS = torch.ops.dreamgnn_mi


@pytest.mark.parametrize("name", ["random", "dups", "empty_rows", "single_row", "E0"])
def test_csr_from_coo_op_bit_exact(dev, name):
    import dream_gnn_amd  # noqa: F401  (registers the ops)

    g = C.load("csr_" + name)
    indptr, indices, eid, flag = OPS.csr_from_coo(C.T(g["row"], dev), C.T(g["col"], dev), int(g["n_rows"]), 0)
    assert np.array_equal(indptr.cpu().numpy(), g["indptr"]) and np.array_equal(indices.cpu().numpy(), g["indices"])
    assert np.array_equal(eid.cpu().numpy(), g["eid"]) and int(flag.item()) == 0


@pytest.mark.parametrize("F", [1, 3, 4, 127, 128, 341, 768])
def test_spmm_csr_op_forward_and_registered_autograd(dev, F):
    """`dreamgnn_mi::spmm_csr(indptr, indices, vals?, X, src_scale?, dst_scale?)`: the functional op
    that stands where layers.py:229-232 / :312 call into DGL / ATen, differentiable w.r.t. X."""
    import dream_gnn_amd  # noqa: F401

    g = C.load("spmm_F%d" % F)
    n_dst, n_src = int(g["n_dst"]), int(g["n_src"])
    dst, src = C.T(g["dst"], dev), C.T(g["src"], dev)
    indptr, indices, eid, _ = OPS.csr_from_coo(dst, src, n_dst, n_src)
    for weighted in (True, False):
        vals = C.T(g["val"], dev)[eid.long()] if weighted else None
        x = C.T(g["X"], dev).requires_grad_(True)
        y = OPS.spmm_csr(indptr, indices, vals, x)
        assert y.requires_grad
        y.backward(C.T(g["dY"], dev))
        tag = "weighted" if weighted else "unit_spmm"
        C.close(y, g["y_" + tag], 1e-5, "y_" + tag)
        C.close(x.grad, g["dx_" + tag], 1e-5, "dx_" + tag)
    # both diagonal scalings fused, against an explicit composition of the op with itself
    rng = np.random.default_rng(F)
    ss = C.T(rng.uniform(0.5, 1.5, n_src).astype(np.float32), dev)
    ds = C.T(rng.uniform(0.5, 1.5, n_dst).astype(np.float32), dev)
    x = C.T(g["X"], dev).requires_grad_(True)
    y = OPS.spmm_csr(indptr, indices, None, x, ss, ds)
    y.backward(C.T(g["dY"], dev))
    x2 = C.T(g["X"], dev).requires_grad_(True)
    y2 = ds[:, None] * OPS.spmm_csr(indptr, indices, None, x2 * ss[:, None])
    y2.backward(C.T(g["dY"], dev))
    C.close(y, y2.detach().cpu().numpy(), 1e-5, "scaled y")
    C.close(x.grad, x2.grad.cpu().numpy(), 1e-5, "scaled dx")


def test_raw_ops_out_variants_and_scratch_reuse(dev):
    import dream_gnn_amd  # noqa: F401
    from dream_gnn_amd import ops

    gen = torch.Generator().manual_seed(0)
    n_dst, n_src, E, F = 400, 300, 20_000, 64
    dst = torch.randint(0, n_dst, (E,), generator=gen, dtype=torch.int32).to(dev)
    src = torch.randint(0, n_src, (E,), generator=gen, dtype=torch.int32).to(dev)
    g = ops.CSRGraph(dst, src, n_dst, n_src)
    X = torch.randn(n_src, F, device=dev)
    a = OPS.spmm_csr_raw(g.indptr, g.indices, None, None, None, X, None, None, None, 0)
    out = torch.empty(n_dst, F, device=dev)
    assert OPS.spmm_csr_out(g.indptr, g.indices, None, None, None, X, None, None, g.plan.buf, g.plan.chunk, out) is None
    assert float((a - out).abs().max()) <= 1e-5 * float(a.abs().max())  # planned: long rows are summed chunk by chunk
    seg, idx, eid, flag = OPS.csr_sliced_from_coo(dst, src, n_dst, n_src, 8)
    b = OPS.spmm_sliced_raw(seg, idx, None, None, None, X, None, None, n_dst, 8)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) and int(flag.item()) == 0
    before = torch.cuda.memory_allocated()
    for _ in range(5):  # scratch (planes / partials) is cached per stream: no allocation growth per call
        OPS.spmm_sliced_out(seg, idx, None, None, None, X, None, None, n_dst, 8, out)
        OPS.spmm_csr_out(g.indptr, g.indices, None, None, None, X, None, None, g.plan.buf, g.plan.chunk, out)
    assert torch.cuda.memory_allocated() == before


def test_ops_reject_bad_arguments_with_runtime_errors(dev):
    import dream_gnn_amd  # noqa: F401

    i32 = lambda *v: torch.tensor(v, dtype=torch.int32, device=dev)
    X = torch.randn(3, 8, device=dev)
    with pytest.raises(RuntimeError, match="must be Int"):
        OPS.spmm_csr(torch.tensor([0, 1], device=dev), i32(0), None, X)
    with pytest.raises(RuntimeError, match="2-D float32"):
        OPS.spmm_csr(i32(0, 1), i32(0), None, X.double())
    with pytest.raises(RuntimeError, match="entries, expected"):
        OPS.spmm_csr(i32(0, 1), i32(0), None, X, torch.ones(2, device=dev))
    with pytest.raises(RuntimeError, match="MI355X only"):
        OPS.spmm_csr_raw(i32(0, 1), i32(0), None, None, None, X.cpu(), None, None, None, 0)
    with pytest.raises(RuntimeError, match="not part of the path"):
        OPS.spmm_csr(i32(0, 1), i32(0), torch.ones(1, device=dev, requires_grad=True), X.requires_grad_(True))
    with pytest.raises(RuntimeError, match="eid"):
        OPS.spmm_csr_raw(i32(0, 1), i32(0), None, None, torch.zeros(1, 8, dtype=torch.int32, device=dev), X, None, None, None, 0)
    with pytest.raises(NotImplementedError):  # no CPU kernel is registered: the dispatcher says so
        OPS.gather_f32(torch.ones(3), torch.zeros(3, dtype=torch.int32))


def test_colsum_rows_ops_match_torch(dev):
    """(f3) dgmi_weighted_colsum_f32 / dgmi_rank_add_f32 through the dispatcher, against the GEMMs they replace."""
    from dream_gnn_amd import ops

    torch.manual_seed(2)
    for n, R, W, B, i0 in ((763, 2, 344, 1, 0), (1256, 2, 256, 2, 0), (37, 3, 5, 3, 2), (5, 1, 128, 1, 0)):
        feat = torch.randn(n * R + B, W, device=dev)
        coef = torch.rand(B, n, device=dev)
        want = (coef.double() @ feat[: n * R].view(n, R, W)[:, i0, :].double())
        keep = feat[: n * R].clone()
        ops.colsum_rows_(feat, coef, n, R, i0)
        assert torch.equal(feat[: n * R], keep)
        assert float((feat[n * R:].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
        again = feat.clone()
        ops.colsum_rows_(again, coef, n, R, i0)
        assert torch.equal(again, feat)  # fixed summation order
        gf = torch.randn(n * R, W, device=dev)
        gs = torch.randn(B, W, device=dev)
        want_g = gf.clone().double()
        want_g.view(n, R, W)[:, i0, :] += coef.double().t() @ gs.double()
        ops.colsum_rows_backward_(gf, coef, gs, n, R, i0)
        assert float((gf.double() - want_g).abs().max()) <= 1e-5 * float(want_g.abs().max())
    with pytest.raises(RuntimeError):
        ops.colsum_rows_(torch.randn(10, 4, device=dev), torch.rand(1, 4, device=dev), 4, 2, 0)  # 4*2 + 1 != 10


def test_scale_rows_matches_torch_and_strided_input(dev):
    """dgmi_scale_rows_f32 (the row-scale pass ahead of XCD-local products): bit-equal to the elementwise product,
    row-strided and unaligned inputs included."""
    from dream_gnn_amd import _lib

    ops_ = _lib.torch_ops
    torch.manual_seed(3)
    for n, F in ((1000, 128), (77, 341), (5, 4), (0, 8)):
        X = torch.randn(n, F, device=dev)
        sc = torch.rand(n, device=dev) + 0.5
        assert torch.equal(ops_.scale_rows(X, sc), X * sc.view(-1, 1))
        wide = torch.randn(n, F + 12, device=dev)
        assert torch.equal(ops_.scale_rows(wide[:, 3:3 + F], sc), wide[:, 3:3 + F] * sc.view(-1, 1))
    with pytest.raises(RuntimeError):
        ops_.scale_rows(torch.randn(4, 8, device=dev), torch.rand(5, device=dev))
