"""BASELINE configs at their OWN sizes on the GPU (configs_untested).

  cfg 1/2  lrssl shape: 763 x 681, every cell a sample, 90 % of them train pairs (467 643 edges per
           direction; relation "0" is 89 % dense, relation "1" holds 2 746 positives), 3 GCMC layers
           at 1024//3 = 341 -> 128 -> 128, FGCN 763/681 -> 768 -> 128 on kNN-4 graphs
           (reference train.py:404-448 defaults)
  cfg 3    Cdataset + Gdataset merged: block-diagonal 1 256 x 722, 411 099 train pairs, 256-d
  cfg 5    rank 0's shard of the 8-rank problems, on one GPU: node-scaled 800k x 400k / 80 M edges
           (410 MB replicated feature table) and edge-scaled 100k x 50k / 80 M edges

The datasets themselves are absent: shapes only.  Per-slice products are checked
against the f64 CPU oracle on EVERY row (465 k edges is cheap on the CPU); module outputs against
the same modules run on the CPU with the oracle patched in as the op backend; the 80 M-edge shards
through size-independent properties plus oracle rows.
"""
import numpy as np
import pytest
import torch

import _cpu_backend

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north_star: fp32 embeddings within 1e-5 relative


def _all_rows_vs_oracle(oracle, indptr, indices, vals, X, ss, ds, y, what):
    ref = oracle.spmm_csr(indptr, indices, vals, X, ss, ds, acc="f64")
    bound = oracle.spmm_csr(indptr, indices, vals, X, ss, ds, acc="abs")  # sum |terms| per element
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= RTOL * bound + 1e-30), "%s: worst element %.2e of its bound" % (what, float((err / (bound + 1e-30)).max()))
    assert err.max() <= RTOL * np.abs(ref).max(), what


@pytest.fixture(scope="module")
def lrssl(dev):
    from dream_gnn_amd import synth

    batch, labels = synth.dataset_shaped_batch([synth.DATASET_SHAPES["lrssl"]], emb=768, k=4, seed=0, device=dev)
    return batch, labels


@pytest.mark.parametrize("F", [344, 341, 128])
def test_lrssl_slices_every_row_every_kernel(oracle, lrssl, dev, F):
    """All four relation slices of the lrssl-shaped encoder graph (two ~465 k-edge 89 %-dense ones,
    two 2.7 k-edge ones), cj/ci fused, F = 341 (layer 0 as the reference sizes it), 344 (the padded
    width the drop-in runs at) and 128 (layers 1-2): every kernel form, every destination row."""
    from dream_gnn_amd import ops

    batch, _ = lrssl
    enc = batch["enc_graph"]
    assert enc.number_of_edges() == 2 * 467_643
    rng = np.random.default_rng(F)
    for can in enc.canonical_etypes:
        rel = enc[can]
        g = rel.csr
        E = g.nnz
        assert E in (464_897, 2_746), E
        X = rng.standard_normal((rel.n_src, F)).astype(np.float32)
        cj = rel.srcdata["cj"].reshape(-1)
        ci = rel.dstdata["ci"].reshape(-1)
        Xd = torch.from_numpy(X).to(dev)
        indptr, indices = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
        # bit-exact CSR indexing against the oracle's stable sort
        ip, ix, ei = oracle.csr_from_coo(rel.dst.cpu().numpy(), rel.src.cpu().numpy(), rel.n_dst)
        assert np.array_equal(indptr, ip) and np.array_equal(indices, ix) and np.array_equal(g.eid.cpu().numpy(), ei)
        forms = {"wave-per-row": ops.spmm_csr_raw(g.indptr, g.indices, None, Xd, cj, ci),
                 "planned": ops.spmm_csr_raw(g.indptr, g.indices, None, Xd, cj, ci, plan=g.plan),
                 "CSRGraph.spmm (the form the modules call)": g.spmm(Xd, cj, ci)}
        if F % 4 == 0:
            forms["xcd-sliced"] = ops.SlicedCSR(rel.dst, rel.src, rel.n_dst, rel.n_src).spmm(Xd, cj, ci)
        for name, y in forms.items():
            _all_rows_vs_oracle(oracle, indptr, indices, None, X, cj.cpu().numpy(), ci.cpu().numpy(), y,
                                "%s %s F=%d" % (can[1], name, F))
        # transpose product (autograd) on the big slices
        W = rng.standard_normal((rel.n_dst, F)).astype(np.float32)
        tp, ti, _ = oracle.csr_from_coo(rel.src.cpu().numpy(), rel.dst.cpu().numpy(), rel.n_src)
        _all_rows_vs_oracle(oracle, tp, ti, None, W, ci.cpu().numpy(), cj.cpu().numpy(),
                            g.spmm_t(torch.from_numpy(W).to(dev), cj, ci), "%s transpose F=%d" % (can[1], F))


@pytest.mark.parametrize("shape", ["lrssl", "C+G"])
@pytest.mark.parametrize("F", [344, 128])
def test_complement_form_every_row_with_and_without_dropout(oracle, lrssl, dev, shape, F):
    """(f3) The relation-fused aggregate in COMPLEMENT form — label-0 relation as
    colsum - complement cells (- its dropped edges under edge dropout: description inverted, live edges
    compacted in-wave) — against the f64 oracle of the PLAIN aggregate diag(ci) sum_r A_r diag(cj_r) X W_r,
    every destination row, both destination types, un-dropped and with the reference's 10 % edge dropout
    (augmentation.py:48-52), forward and transpose (through autograd, so that the virtual source's gradient
    flows back through the column sum)."""
    from dream_gnn_amd import graph as G, ops, synth

    if shape == "lrssl":
        enc = lrssl[0]["enc_graph"]
    else:
        drug, dis, labels, nd, ns = synth.dataset_shaped_pairs([synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]])
        enc = G.build_enc_graph(drug, dis, labels, nd, ns, device=dev).int()
    rng = np.random.default_rng(F)
    torch.manual_seed(5)
    child = G.random_edge_dropout(enc, 0.1)
    for g_, who in ((enc, "un-dropped"), (child, "10 % edge dropout")):
        for nt in ("disease", "drug"):
            full, cans = g_.fused_relations(nt)
            comp = g_.fused_relations_complement(nt)
            assert comp is not None and comp[1] == cans
            ccsr, _, i0, blockmat = comp
            R = len(cans)
            n_src = full.n_src // R
            cells = n_src * full.n_dst
            if who == "un-dropped":
                assert ccsr.nnz < 0.25 * full.nnz, (ccsr.nnz, full.nnz)  # ~8x fewer edges walked and gathered
            X = rng.standard_normal((full.n_src, F)).astype(np.float32)
            ss = rng.uniform(0.5, 1.5, full.n_src).astype(np.float32)
            ds = rng.uniform(0.5, 1.5, full.n_dst).astype(np.float32)
            W = rng.standard_normal((full.n_dst, F)).astype(np.float32)
            # oracle of the plain aggregate over the SURVIVING edges of the fused edge list
            S_ = full._S
            m = full.keep_mask()
            keep = np.ones(full.nnz, bool) if m is None else m.cpu().numpy().astype(bool)
            dst, src = S_.dst.cpu().numpy()[keep], S_.src.cpu().numpy()[keep]
            ip, ix, _ = oracle.csr_from_coo(dst, src, full.n_dst)
            tp, ti, _ = oracle.csr_from_coo(src, dst, full.n_src)
            t = lambda a: torch.from_numpy(a).to(dev)
            xa = t(X).requires_grad_(True)
            assert blockmat.shape == ((1 if shape == "lrssl" else 2), n_src)  # C+G merged: one column sum per dataset block
            s_row = blockmat @ (t(ss).view(-1, R)[:, i0:i0 + 1] * xa.view(n_src, R, F)[:, i0])
            y = ops.spmm_csr(ccsr, torch.cat([xa, s_row]), torch.cat([t(ss), torch.ones(blockmat.shape[0], device=dev)]), t(ds))
            # the identity subtracts ~10 % of the terms from the full column sum: the error bound is the plain sum's
            # plus the subtracted part's, i.e. at most the bound over ALL cells of relation i0 — use the oracle's
            # sum |terms| of the plain product scaled by cells / kept (>= every term that entered)
            ref = oracle.spmm_csr(ip, ix, None, X, ss, ds, acc="f64")
            bound = oracle.spmm_csr(ip, ix, None, X, ss, ds, acc="abs") * (cells / max(1, keep.sum() // 1))
            err = np.abs(y.detach().cpu().numpy().astype(np.float64) - ref)
            assert np.all(err <= RTOL * bound + 1e-30), "%s %s ->%s F=%d: %.2e of the bound" % (shape, who, nt, F, float((err / (bound + 1e-30)).max()))
            assert err.max() <= RTOL * np.abs(ref).max(), (shape, who, nt, F, err.max(), np.abs(ref).max())
            y.backward(t(W))
            ref_t = oracle.spmm_csr(tp, ti, None, W, ds, ss, acc="f64")
            err_t = np.abs(xa.grad.cpu().numpy().astype(np.float64) - ref_t)
            assert err_t.max() <= RTOL * np.abs(ref_t).max(), (shape, who, nt, F, "transpose", err_t.max(), np.abs(ref_t).max())


def test_lrssl_knn4_adjacencies_every_row(oracle, lrssl, dev):
    """The four kNN-4 FGCN graphs (nnz <= 9 N) at the widths the model runs them: 768 and 128."""
    from dream_gnn_amd import layers as L

    batch, _ = lrssl
    rng = np.random.default_rng(3)
    for name in ("drug_graph", "disease_graph", "drug_feature_graph", "disease_feature_graph"):
        adj = batch[name]
        g = L.adjacency_csr(adj)
        n = adj.shape[0]
        assert n in (763, 681) and g.nnz <= 9 * n
        for F in (768, 128):
            X = rng.standard_normal((n, F)).astype(np.float32)
            y = g.spmm(torch.from_numpy(X).to(dev))
            _all_rows_vs_oracle(oracle, g.indptr.cpu().numpy(), g.indices.cpu().numpy(), g.vals.cpu().numpy(), X,
                                None, None, y, "%s F=%d" % (name, F))
            # and against the call being replaced, on the same device: th.spmm (layers.py:312)
            ref = torch.spmm(adj, torch.from_numpy(X).to(dev))
            assert float((y - ref).abs().max()) <= RTOL * float(ref.abs().max())


def _model_grads(device, blocks, out_units, seed, complement, state=None, f64=False):
    """(loss, logits, parameter gradients, state_dict) of one forward + backward of the full Net, no stochastic
    layers; ``f64``: everything cast to double (CPU, float64 SpMM backend)."""
    from dream_gnn_amd import harness as H, layers as L, model as M, synth

    old = L.GCMCLayer.complement_form
    L.GCMCLayer.complement_form = complement
    try:
        batch, labels = synth.dataset_shaped_batch(blocks, emb=768, k=4, seed=seed, device=device)
        args = synth.net_args(out_units=out_units, n_drug=batch["n_drug"], n_dis=batch["n_dis"], dropout=0.0, attention_dropout=0.0)
        torch.manual_seed(7)
        net = M.Net(args)
        if state is not None:
            net.load_state_dict(state)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        if f64:
            net = net.double()
            batch = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in batch.items()}
            labels = labels.double()
        net = net.to(device).train()
        loss, pred = H.forward_loss(net, batch, labels, beta=0.1)
        loss.backward()
        return (float(loss.detach()), pred.detach().cpu().double(),
                {k: p.grad.detach().cpu().double() for k, p in net.named_parameters() if p.grad is not None}, state)
    finally:
        L.GCMCLayer.complement_form = old


@pytest.mark.parametrize("blocks_name,out_units", [("lrssl", 128), ("C+G", 256)])
def test_full_model_gradients_of_both_forms_against_an_f64_evaluation(oracle, dev, blocks_name, out_units):
    """the plain fused form and the COMPLEMENT form (f3) are two fp32 evaluations of the same model;
    compared with each other (HIP vs CPU-oracle path) the complement form looked 60x worse on the parameters whose
    gradient is a near-total cancellation over all nodes (1.7e-2 vs 2.7e-4 of the tensor's max on TGCN.0.basis).
    Against an evaluation of the whole model in float64 (CPU, torch double, float64 SpMM) both forms are held to the SAME
    bounds — at lrssl shape 1e-4 of the tensor's maximum for every GCMC-channel weight (TGCN.*.att / basis / ufc.weight:
    where the forms differ) and 1e-3 for everything else (bias gradients are cancelling sums over every node / pair;
    FGCN.gc1 sits behind a relu whose sign flips under fp32 rounding: 7e-4 already between MKL fp32 and float64 on the
    CPU) — and, tensor by tensor, the complement form may not be worse than the plain one by more than rounding noise
    (1.5x, or 1e-4)."""
    from dream_gnn_amd import synth

    blocks = [synth.DATASET_SHAPES["lrssl"]] if blocks_name == "lrssl" else [synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]]
    seed = 0 if blocks_name == "lrssl" else 1
    with _cpu_backend.patched(f64=True):
        l64, p64, g64, state = _model_grads(torch.device("cpu"), blocks, out_units, seed, complement=False, f64=True)
    worst = {}
    for complement in (False, True):
        l, p, g, _ = _model_grads(dev, blocks, out_units, seed, complement=complement, state=state)
        assert abs(l - l64) <= 1e-5 * max(1.0, abs(l64))
        assert float((p - p64).abs().max()) <= 1e-4 * float(p64.abs().max())
        assert set(g) == set(g64)
        worst[complement] = {k: float((g[k] - g64[k]).abs().max()) / max(float(g64[k].abs().max()), 1e-30) for k in g64}
    top = lambda d: sorted(((round(e, 7), k) for k, e in d.items() if k.startswith("TGCN.")), reverse=True)[:4]
    print("%s: worst TGCN gradient errors vs f64 — plain %s; complement %s" % (blocks_name, top(worst[False]), top(worst[True])))
    # lrssl shape: 1e-4 on the GCMC-channel weights (measured 4.2e-5 / 4.4e-5), 1e-3 elsewhere.  Merged C+G shape (two
    # blocks, 256-d): fp32 itself is further from float64 there — the PLAIN form errs by 3.6e-3 on TGCN.1.basis (a sum over
    # 1 978 nodes x 256 columns that cancels almost completely) and 6e-4 on TGCN.0.ufc.bias, the complement form by the same
    # 3.6e-3 / 6e-4 — so the absolute bound is 5e-3 there and the statement that matters is the relative one below.
    tight, loose = (1e-4, 1e-3) if blocks_name == "lrssl" else (5e-3, 5e-3)
    for complement in (False, True):
        for k, e in worst[complement].items():
            is_weight = k.startswith("TGCN.") and not k.endswith(".bias")
            assert e <= (tight if is_weight else loose), (("complement" if complement else "plain"), k, e)
    for k, e in worst[True].items():  # the complement form is as accurate as the plain one, tensor by tensor
        assert e <= max(1.5 * worst[False][k], 1e-4), ("complement form worse than the plain one", k, e, worst[False][k])


def _module_parity_runs(dev, blocks, out_units, seed, res, H, M, synth):
    for where in ("gpu", "cpu"):

        device = dev if where == "gpu" else torch.device("cpu")
        ctx = _cpu_backend.patched() if where == "cpu" else None
        if ctx is not None:
            ctx.__enter__()
        try:
            batch, labels = synth.dataset_shaped_batch(blocks, emb=768, k=4, seed=seed, device=device)
            args = synth.net_args(out_units=out_units, n_drug=batch["n_drug"], n_dis=batch["n_dis"], dropout=0.0,
                                  attention_dropout=0.0)
            torch.manual_seed(7)
            net = M.Net(args)
            if "state" in res:
                net.load_state_dict(res["state"])
            res.setdefault("state", {k: v.clone() for k, v in net.state_dict().items()})
            net = net.to(device).train()
            loss, pred = H.forward_loss(net, batch, labels, beta=0.1)
            loss.backward()
            res[where] = (float(loss.detach()), pred.detach().cpu(),
                          {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None})
            if where == "gpu":  # the layer-0 width the reference derives: 1024 // 3
                assert net.TGCN[0].msg_units == 341 and net.TGCN[1].msg_units == out_units
                fused = batch["enc_graph"].__dict__.get("_fused", {})
                assert set(fused) == {"drug", "disease"} and all(v is not None for v in fused.values())
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)


def _module_parity(dev, blocks, out_units, seed, complement=False):
    """Net forward + loss + backward on the HIP path vs the same modules on the CPU with the
    oracle as op backend, identical parameters and inputs, no stochastic layers."""
    from dream_gnn_amd import harness as H, layers as L, model as M, synth

    res = {}
    default_form = L.GCMCLayer.complement_form
    L.GCMCLayer.complement_form = complement
    try:
        _module_parity_runs(dev, blocks, out_units, seed, res, H, M, synth)
    finally:
        L.GCMCLayer.complement_form = default_form  # whatever the outcome
    (gl, gp, gg), (cl, cp, cg) = res["gpu"], res["cpu"]
    assert abs(gl - cl) <= 1e-5 * max(1.0, abs(cl)), (gl, cl)
    assert float((gp - cp).abs().max()) <= 1e-4 * float(cp.abs().max()), "logits"
    assert set(gg) == set(cg)
    # parameter gradients pass through 3 layers of fp32 GEMMs whose reduction order differs between
    # hipBLASLt and MKL; bias gradients are cancelling sums over every node / pair (worst measured:
    # 2.7e-4 of the tensor's max on TGCN.2.ufc.bias at the merged shape)
    # (The COMPLEMENT form is not compared here: two fp32 evaluations that round one shared column sum differently sit
    # 1.7e-2 apart on TGCN.0.basis while each is within 4.4e-5 of an f64 evaluation — both forms are held to the same
    # bounds against float64 in test_full_model_gradients_of_both_forms_against_an_f64_evaluation instead.)
    tol = 1e-3
    for k in cg:
        scale = float(cg[k].abs().max())
        assert float((gg[k] - cg[k]).abs().max()) <= tol * scale + 1e-9, (k, scale)


def test_cfg2_lrssl_full_model_hip_vs_cpu_oracle_path(oracle, dev):
    from dream_gnn_amd import synth

    _module_parity(dev, [synth.DATASET_SHAPES["lrssl"]], 128, seed=0)


def test_cfg3_c_plus_g_merged_full_model_hip_vs_cpu_oracle_path(oracle, dev):
    from dream_gnn_amd import synth

    _module_parity(dev, [synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]], 256, seed=1)


def test_cfg3_slices_every_row(oracle, dev):
    """The merged block-diagonal encoder graph at F = 256 (layers 1-2 of config 3)."""
    from dream_gnn_amd import synth

    batch, _ = synth.dataset_shaped_batch([synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]],
                                          emb=8, k=4, seed=1, device=dev)
    enc = batch["enc_graph"]
    assert (batch["n_drug"], batch["n_dis"]) == (1256, 722) and enc.number_of_edges() == 2 * 411_099
    rng = np.random.default_rng(9)
    for can in enc.canonical_etypes:
        rel = enc[can]
        # block-diagonal: no pair crosses the two datasets
        cross = ((rel.src < (663 if can[0] == "drug" else 409)) != (rel.dst < (409 if can[0] == "drug" else 663)))
        assert not bool(cross.any())
        X = rng.standard_normal((rel.n_src, 256)).astype(np.float32)
        cj, ci = rel.srcdata["cj"].reshape(-1), rel.dstdata["ci"].reshape(-1)
        g = rel.csr
        _all_rows_vs_oracle(oracle, g.indptr.cpu().numpy(), g.indices.cpu().numpy(), None, X, cj.cpu().numpy(),
                            ci.cpu().numpy(), g.spmm(torch.from_numpy(X).to(dev), cj, ci), "%s F=256" % can[1])


# ---------------------------------------------------------------------------------------------
# config 5: one rank's shard of the 8-rank problems
# ---------------------------------------------------------------------------------------------
def _shard_checks(oracle, dev, nd, ns, E, world=8, F=128):
    from dream_gnn_amd import shard as S, synth

    drug, dis = synth.bipartite_edges(nd, ns, E, seed=4, device=dev)
    deg_in = torch.bincount(dis.long(), minlength=ns)
    bounds = S.balanced_row_bounds(deg_in, world)
    sh = S.RowShard(dis, drug, ns, nd, bounds, 0)  # drug -> disease, rank 0's rows and all their in-edges
    assert abs(sh.nnz - E // world) <= 0.01 * E / world  # nnz-balanced
    g = sh.local
    gen = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn(nd, F, generator=gen, device=dev)
    cj = synth.degree_norm(drug, nd)
    ci = synth.degree_norm(dis, ns)
    del drug
    y = sh.spmm_local(X, cj, ci)
    assert torch.equal(y, sh.spmm_local(X, cj, ci))  # reproducible
    # A @ ones = in-degree exactly
    deg = (g.indptr[1:] - g.indptr[:-1]).float()
    assert torch.equal(g.spmm(torch.ones(nd, 8, device=dev))[:, 0], deg)
    assert torch.equal(deg.long(), deg_in[sh.lo:sh.hi])
    # adjoint identity with both scalings on the local block
    ci_loc = ci[sh.lo:sh.hi].contiguous()
    W = torch.randn(sh.hi - sh.lo, F, generator=gen, device=dev)
    lhs = (y.double() * W.double()).sum()
    rhs = (X.double() * g.spmm_t(W, cj, ci_loc).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float((y.double().abs() * W.double().abs()).sum())
    # linearity
    Z = torch.randn(nd, F, generator=gen, device=dev)
    lin = sh.spmm_local(2.0 * X - 0.5 * Z, cj, ci)
    ref = 2.0 * y - 0.5 * sh.spmm_local(Z, cj, ci)
    assert float((lin - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    del Z, lin, ref
    # f64 oracle on sampled rows of the shard
    indptr, indices = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
    Xn, cjn, cin = X.cpu().numpy(), cj.cpu().numpy(), ci_loc.cpu().numpy()
    rows = [0, sh.hi - sh.lo - 1, int(deg.argmax())] + np.random.default_rng(0).integers(0, sh.hi - sh.lo, 30).tolist()
    for r in rows:
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        ref = oracle.spmm_csr(np.array([0, hi - lo], np.int32), indices[lo:hi], None, Xn, cjn, cin[r:r + 1], acc="f64")[0]
        got = y[r].cpu().numpy().astype(np.float64)
        assert np.abs(got - ref).max() <= RTOL * max(np.abs(ref).max(), 1e-30), "row %d" % r
    return sh


def test_cfg5_node_scaled_rank0_shard(oracle, dev):
    """800k x 400k / 80 M edges (SURVEY §8d's cfg 5): rank 0 of 8 owns ~50k disease rows and their
    ~10 M in-edges; the replicated feature table is 410 MB (no L2 slicing possible)."""
    sh = _shard_checks(oracle, dev, 800_000, 400_000, 80_000_000)
    assert sh.local.n_src == 800_000 and 49_000 <= sh.hi - sh.lo <= 51_000


def test_cfg5_edge_scaled_rank0_shard(oracle, dev):
    """100k x 50k / 80 M edges (cfg-4 nodes, 8x the edges): rank 0 owns ~6 250 rows of degree ~1 600."""
    sh = _shard_checks(oracle, dev, 100_000, 50_000, 80_000_000)
    assert sh.local.n_src == 100_000 and 6_000 <= sh.hi - sh.lo <= 6_500
