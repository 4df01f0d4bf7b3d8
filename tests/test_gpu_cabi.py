"""The C ABI driven from Python on the GPU exactly as INTEGRATION.md §2 tells a maintainer to: raw
``data_ptr()``s, torch's current stream, ctypes — no dream_gnn_amd import.  The stub is not restated here: the
code block of INTEGRATION.md §2 is extracted and executed verbatim, so the document cannot drift from what runs.
Interface: include/dgmi.h (dgmi_csr_from_coo_i32 two-call workspace protocol, dgmi_spmm_csr_f32,
dgmi_csr_sliced_from_coo_i32 + dgmi_spmm_sliced_f32); reference call sites: layers.py:224-234, :312."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-5


@pytest.fixture(scope="module")
def stub(dev):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# dgmi_binding\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md §2 lost its dgmi_binding.py block"
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)  # the stub opens dream_gnn_amd/libdgmi.so relative to the repository root
    try:
        exec(compile(m.group(1), "INTEGRATION.md:dgmi_binding.py", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def _vs_oracle(oracle, y, ip, ix, vals, X, ss, ds, what):
    ref = oracle.spmm_csr(ip, ix, vals, X, ss, ds, acc="f64")
    bound = oracle.spmm_csr(ip, ix, vals, X, ss, ds, acc="abs")
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= RTOL * bound + 1e-30), what


def test_integration_stub_on_the_golden_spmm_fixture(oracle, stub, dev):
    """tests/golden/spmm_F128.npz (produced by the reference's own th.spmm): COO -> CSR through the two-call
    workspace protocol, bit-exact against the oracle's stable sort; the unweighted and the weighted product."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "spmm_F128.npz"))
    dst, src = g["dst"].astype(np.int32), g["src"].astype(np.int32)
    n_dst, n_src = int(g["n_dst"]), int(g["n_src"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    indptr, indices, eid = stub["csr_from_coo"](t(dst), t(src), n_dst, n_src)
    ip, ix, ei = oracle.csr_from_coo(dst, src, n_dst)
    assert np.array_equal(indptr.cpu().numpy(), ip) and np.array_equal(indices.cpu().numpy(), ix) and np.array_equal(eid.cpu().numpy(), ei)
    X = g["X"].astype(np.float32)
    y_unit = stub["copy_u_sum"](indptr, indices, t(X))
    assert float((y_unit.cpu() - torch.from_numpy(g["y_unit_spmm"])).abs().max()) <= RTOL * float(np.abs(g["y_unit_spmm"]).max())
    vals = g["val"].astype(np.float32)[ei]
    y_w = stub["copy_u_sum"](indptr, indices, t(X), vals=t(vals))
    assert float((y_w.cpu() - torch.from_numpy(g["y_weighted"])).abs().max()) <= RTOL * float(np.abs(g["y_weighted"]).max())
    _vs_oracle(oracle, y_w, ip, ix, vals, X, None, None, "weighted")


@pytest.mark.parametrize("F", [128, 344])
def test_integration_stub_on_a_random_graph_every_entry_point(oracle, stub, dev, F):
    """7 000 random edges (duplicates, empty rows), both scalings, a row-strided X: wave-per-row entry point and the
    XCD-sliced pair, each against the f64 oracle on every element."""
    rng = np.random.default_rng(F)
    n_dst, n_src, E = 300, 450, 7000
    dst = rng.integers(0, n_dst, E).astype(np.int32)
    dst[dst == 5] = 6  # an empty destination row
    src = rng.integers(0, n_src, E).astype(np.int32)
    X = rng.standard_normal((n_src, F)).astype(np.float32)
    cj = rng.uniform(0.5, 1.5, n_src).astype(np.float32)
    ci = rng.uniform(0.5, 1.5, n_dst).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    indptr, indices, eid = stub["csr_from_coo"](t(dst), t(src), n_dst, n_src)
    ip, ix, ei = oracle.csr_from_coo(dst, src, n_dst)
    assert np.array_equal(indptr.cpu().numpy(), ip) and np.array_equal(indices.cpu().numpy(), ix) and np.array_equal(eid.cpu().numpy(), ei)
    wide = torch.zeros(n_src, F + 8, device=dev)
    wide[:, :F] = t(X)
    for name, x in (("contiguous", t(X)), ("row-strided view", wide[:, :F])):
        y = stub["copy_u_sum"](indptr, indices, x, t(cj), t(ci))
        _vs_oracle(oracle, y, ip, ix, None, X, cj, ci, name)
        assert float(y[5].abs().max()) == 0.0
        y_s = stub["copy_u_sum_xcd_sliced"](t(dst), t(src), n_dst, n_src, x, t(cj), t(ci))
        _vs_oracle(oracle, y_s, ip, ix, None, X, cj, ci, name + ", xcd-sliced")
    # (r4) the per-iteration edge dropout of train.py:267 through the same raw entry points: select a subset, compact the
    # CSR — bit-identical to the CSR of the kept edge list — and run the plain product on it
    ipk, ixk, desc = stub["edge_dropout"](indptr, indices, eid, 0.25, 4242)
    mask = oracle.keep_mask(desc.cpu().numpy(), E).astype(bool)
    assert int(mask.sum()) == max(1, int(E * 0.75)) and np.array_equal(desc.cpu().numpy(), oracle.random_subset_select(E, int(mask.sum()), 4242, 0))
    kp, ki, _ = oracle.csr_from_coo(dst[mask], src[mask], n_dst)
    assert np.array_equal(ipk.cpu().numpy(), kp) and np.array_equal(ixk.cpu().numpy()[: ki.size], ki)
    y_k = stub["copy_u_sum"](ipk, ixk, t(X), t(cj), t(ci))
    _vs_oracle(oracle, y_k, kp, ki, None, X, cj, ci, "edge-dropped, compacted CSR")
    # errors come back as status codes, as the header says: ld < F
    with pytest.raises(RuntimeError, match="invalid argument"):
        stub["check"](stub["lib"].dgmi_spmm_csr_f32(indptr.data_ptr(), indices.data_ptr(), None, None, None, 0, wide.data_ptr(), 4,
                                                  None, None, y.data_ptr(), F, n_dst, n_src, F, 0, 0.0, None, 0, 1.0, None))
