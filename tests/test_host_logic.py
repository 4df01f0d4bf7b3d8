"""Host-side logic of the drop-in modules on the build box (no GPU): the three ops are
replaced, explicitly and only here, by the oracle-backed functions of tests/_cpu_backend.py;
expected values are the golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

import _cases as C
import _cpu_backend

CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def _backend(oracle):
    with _cpu_backend.patched():
        yield


@pytest.mark.parametrize("name", ["random", "dups", "empty_rows", "single_row", "E0"])
def test_csr(name):
    C.case_csr(CPU, name)


@pytest.mark.parametrize("F", [1, 3, 4, 127, 128, 341, 768])
def test_spmm_autograd(F):
    C.case_spmm(CPU, F)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gcmc_graph_conv(mode):
    C.case_gcmc_conv(CPU, mode)


@pytest.mark.parametrize("fuse,complement", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("name", ["shared_ini", "shared_noini", "unshared", "shareflag_dimdiff"])
def test_gcmc_layer(name, fuse, complement):
    """fuse + complement: the near-complete label-0 relation as colsum - complement (f3) — same
    outputs and gradients as the reference's per-slice run."""
    C.case_gcmc_layer(CPU, name, fuse, complement=complement)


@pytest.mark.parametrize("fuse", [True, False])
def test_gcmc_layer_dropout_draw_order(fuse):
    """Dropout on (p = 0.3): the reference run drew one (N_src, 1) mask per relation, in the graph's
    canonical (sorted, DGL-style) relation order, then the two layer-level masks.  On the CPU the
    torch RNG stream is the reference's, so equal outputs mean equal draw order and placement —
    in the fused path and in the per-slice path alike.  (Order is relative to the stand-in's
    HeteroGraphConv; DGL itself is absent — parity unpinned.)"""
    C.case_gcmc_layer(CPU, "shared_dropout", fuse, dropout_rate=0.3, seed=31337)


@pytest.mark.parametrize("name", ["both", "simonly"])
def test_fgcn(name):
    C.case_fgcn(CPU, name)


def test_graphconv_nobias():
    C.case_graphconv_nobias(CPU)


@pytest.mark.parametrize("fuse_decoder", [True, False])
def test_net_forward_loss_grads(fuse_decoder):
    C.case_net(CPU, fuse_decoder)


@pytest.mark.parametrize("symm", [1, 0])
def test_enc_graph_format(symm):
    C.case_encgraph(CPU, symm)


def test_external_and_own_weight_conflict():
    from dream_gnn_amd import graph as G, layers as L

    hg = G.HeteroGraph({("drug", "0", "disease"): (torch.tensor([0, 1]), torch.tensor([1, 0]))},
                       {"drug": 2, "disease": 2}).int()
    hg.nodes["drug"].data["cj"] = torch.ones(2, 1)
    hg.nodes["disease"].data["ci"] = torch.ones(2, 1)
    conv = L.GCMCGraphConv(4, 4, weight=True)
    with pytest.raises(L.DGMIError):  # layers.py:214-216
        conv(hg["0"], torch.randn(2, 4), weight=torch.randn(4, 4))


def test_dot_or_identity_three_column_branch():
    from dream_gnn_amd.layers import dot_or_identity

    B = torch.arange(20.0).view(5, 4)
    A = torch.tensor([[0, 1, 2], [4, 4, 3]])
    out = dot_or_identity(A, B)  # layers.py:385-389
    assert out.shape == (2, 12) and torch.equal(out[1, :4], B[4]) and torch.equal(out[1, 8:], B[3])
    assert dot_or_identity(None, B) is B


def test_state_dict_keys_match_reference_layout():
    from dream_gnn_amd import layers as L

    shared = L.GCMCLayer([0, 1], 12, 12, 24, 6, agg="sum", share_user_item_param=True)
    assert sorted(shared.state_dict()) == ["att", "basis", "ifc.bias", "ifc.weight", "ufc.bias", "ufc.weight"]
    assert shared.ifc is shared.ufc and shared.msg_units == 8 and shared.W_r is not None
    own = L.GCMCLayer([0, 1], 12, 10, 24, 6, agg="sum", share_user_item_param=False)
    keys = set(own.state_dict())
    assert {"conv.mods.0.weight", "conv.mods.rev-0.weight", "conv.mods.1.weight", "conv.mods.rev-1.weight",
            "att", "basis", "ifc.weight", "ufc.weight"} <= keys
    assert own.W_r is None and own.conv.mods["rev-0"].weight.shape == (10, 8)
    with pytest.raises(AssertionError):  # layers.py:53
        L.GCMCLayer([0, 1], 12, 12, 25, 6, agg="stack")
    f = L.FGCN(7, 5, 16, 8, 0.1)
    assert "FGCN_drug.gc1.weight" in f.state_dict() and "disease_fusion.bias" in f.state_dict()


def test_adjacency_cache_is_per_tensor_and_version():
    from dream_gnn_amd import layers as L

    idx = torch.tensor([[0, 1, 1], [1, 0, 1]])
    adj = torch.sparse_coo_tensor(idx, torch.tensor([1.0, 2.0, 3.0]), (2, 2))
    g1 = L.adjacency_csr(adj)
    assert L.adjacency_csr(adj) is g1
    adj2 = torch.sparse_coo_tensor(idx, torch.tensor([1.0, 2.0, 3.0]), (2, 2))
    assert L.adjacency_csr(adj2) is not g1
    assert L.adjacency_csr(g1) is g1


def test_uncoalesced_duplicates_sum_like_torch_spmm():
    from dream_gnn_amd import layers as L

    idx = torch.tensor([[0, 0, 0, 2], [1, 1, 1, 0]])
    val = torch.tensor([1.0, 2.0, 4.0, 5.0])
    adj = torch.sparse_coo_tensor(idx, val, (3, 2))
    x = torch.randn(2, 6)
    gc = L.GraphConvolution(6, 6, bias=False)
    gc.weight.data = torch.eye(6)
    assert torch.allclose(gc(x, adj), torch.spmm(adj, x), atol=1e-6)


def test_metric_lines_match_evaluation_py():
    from dream_gnn_amd.harness import auroc_aupr

    g = C.load("metrics")
    for i in range(3):
        auroc, aupr = auroc_aupr(g["true%d" % i], g["score%d" % i])
        assert abs(auroc - g["auroc"][i]) < 1e-12 and abs(aupr - g["aupr"][i]) < 1e-12


def test_common_loss_matches_utils_py():
    from dream_gnn_amd.model import common_loss

    g = C.load("common_loss")
    got = common_loss(torch.from_numpy(g["e1"]), torch.from_numpy(g["e2"]))
    assert abs(float(got) - float(g["loss"])) < 1e-7


def test_train_step_structure_on_cpu_backend():
    """augmentation -> forward -> loss -> backward -> clip -> Adam (train.py:249-300) runs through
    the drop-in modules; edge dropout rebuilds the graphs every step."""
    from dream_gnn_amd import graph as G, harness as H, model as M

    g = C.load("net_mini")
    nd, ns = int(g["n_drug"]), int(g["n_dis"])
    args = C.net_args(g)
    args.dropout = 0.1
    torch.manual_seed(0)
    net = C.load_sd(M.Net(args), g, CPU)
    batch = {"enc_graph": C.build_enc(g, CPU),
             "dec_graph": G.build_dec_graph(torch.from_numpy(g["dec_src"]), torch.from_numpy(g["dec_dst"]), nd, ns).int(),
             "drug_graph": C.sparse(g, "drug_graph", nd, nd, CPU), "disease_graph": C.sparse(g, "dis_graph", ns, ns, CPU),
             "drug_feature_graph": C.sparse(g, "drug_fg", nd, nd, CPU), "disease_feature_graph": C.sparse(g, "dis_fg", ns, ns, CPU),
             "drug_feat": torch.from_numpy(g["drug_feat"]), "disease_feat": torch.from_numpy(g["dis_feat"]),
             "drug_sim_feat": torch.from_numpy(g["drug_sim"]), "disease_sim_feat": torch.from_numpy(g["dis_sim"])}
    labels = torch.from_numpy(g["values"])
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)  # train.py:217
    losses = [float(H.train_step(net, opt, batch, labels, beta=0.1)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    aug = H.augment(batch)
    assert aug["enc_graph"].number_of_edges("0") == max(1, int(batch["enc_graph"].number_of_edges("0") * 0.9))
    nnz = batch["drug_graph"]._values().shape[0]  # harness.augment drops sparse adjacencies as masked CSR views
    assert int((aug["drug_graph"].vals != 0).sum()) == max(1, int(nnz * 0.9)) and aug["drug_graph"].nnz == nnz
    sp_drop = G.random_edge_dropout_sparse(batch["drug_graph"], 0.1)  # the reference's return type is still available
    assert sp_drop.is_sparse and sp_drop._values().shape[0] == max(1, int(nnz * 0.9))
    assert aug["dec_graph"] is batch["dec_graph"] and not torch.equal(aug["drug_feat"], batch["drug_feat"])
    auroc, aupr = H.evaluate(net, batch, labels)
    assert 0.0 <= auroc <= 1.0 and 0.0 <= aupr <= 1.0


@pytest.mark.parametrize("name", ["n30_k4", "n12_k20"])
def test_similarity_graph_builder(name):
    C.case_similarity_graph(CPU, name)


def test_dgl_shaped_graph_is_accepted():
    """A graph exposing DGL's accessor surface (here: the fixture generator's stand-in, DGL itself
    being absent) goes straight into GCMCLayer and gives the reference's output."""
    import sys

    sys.path.insert(0, C.GOLD)
    import dgl_standin

    from dream_gnn_amd import graph as G, layers as L

    g = C.load("gcmc_layer_shared_ini")
    ours = C.build_enc(g, CPU)
    foreign = dgl_standin.DGLGraph({can: ours.edges(etype=can) for can in ours.canonical_etypes},
                                   {"drug": int(g["n_drug"]), "disease": int(g["n_dis"])})
    for nt in ("drug", "disease"):
        foreign.nodes[nt].data.update(ours.nodes[nt].data)
    conv = G.from_dgl(foreign)
    assert conv.canonical_etypes == ours.canonical_etypes and torch.equal(conv.nodes["drug"].data["ci"], ours.nodes["drug"].data["ci"])
    uin, min_, msg, out, ini, share = [int(v) for v in g["cfg"]]
    layer = C.load_sd(L.GCMCLayer([0, 1], uin, min_, msg, out, dropout_rate=0.0, agg="sum",
                                  agg_act=L.get_activation("leaky"), ini=bool(ini), share_user_item_param=bool(share)), g, CPU)
    o_drug, o_dis = layer(foreign, torch.from_numpy(g["drug"]), torch.from_numpy(g["dis"]))
    C.close(o_drug, g["o_drug"], 1e-5, "o_drug")
    C.close(o_dis, g["o_dis"], 1e-5, "o_dis")


@pytest.mark.parametrize("selection", ["randperm", "select"])
def test_edge_dropout_is_a_mask_view_with_the_same_product(selection):
    """augmentation.py:13-124 without re-sorting: the dropped graph's product equals the product
    over a CSR rebuilt from its (materialised) kept edge lists; same for the sparse adjacency.
    ``select``: the subset is an 8-word description evaluated per edge (CSRGraph.dropped);
    ``randperm``: the reference's literal permutation prefix as a 0/1 value mask."""
    from dream_gnn_amd import graph as G, layers as L, ops

    g = C.load("gcmc_layer_shared_ini")
    enc = C.build_enc(g, CPU)
    torch.manual_seed(3)
    child = G.random_edge_dropout(enc, 0.3, selection=selection)
    assert all((child[c].desc is not None) == (selection == "select") for c in enc.canonical_etypes)
    X = {"drug": torch.randn(int(g["n_drug"]), 8), "disease": torch.randn(int(g["n_dis"]), 8)}
    for can in enc.canonical_etypes:
        rel, base = child[can], enc[can]
        assert isinstance(rel, G.DroppedRelation) and rel.number_of_edges() == max(1, int(base.number_of_edges() * 0.7))
        assert rel.csr.indptr is base.csr.indptr  # structure shared, not rebuilt
        rebuilt = ops.CSRGraph(rel.dst, rel.src, rel.n_dst, rel.n_src)
        x = X[can[0]]
        assert torch.allclose(rel.csr.spmm(x), rebuilt.spmm(x), atol=1e-5)
        w = torch.randn(rel.n_dst, 8)
        assert torch.allclose(rel.csr.spmm_t(w), rebuilt.spmm_t(w), atol=1e-5)
    fused_child, fused_parent = child.fused_relations("disease"), enc.fused_relations("disease")
    assert fused_child[0].indptr is fused_parent[0].indptr
    assert (fused_child[0].vals is not None) == (selection == "randperm")
    # the fused (two-relation) view drops exactly the union of the two per-relation subsets
    cans = fused_child[1]
    want = torch.cat([child[c].keep_mask() for c in cans])
    got = fused_child[0].keep_mask() if selection == "select" else fused_child[0]._coo_vals
    assert torch.equal(got, want) and int(want.sum()) == sum(child[c].number_of_edges() for c in cans)
    xf = torch.randn(fused_parent[0].n_src, 8)
    ref = sum(child[c].csr.spmm(xf.view(-1, len(cans), 8)[:, i].contiguous()) for i, c in enumerate(cans))
    assert torch.allclose(fused_child[0].spmm(xf), ref, atol=1e-5)
    # complement form of the same aggregate (f3): the near-complete relation i0 enters as colsum - complement cells -
    # its DROPPED edges (its description inverted), the other relation with its description as is
    for g_, who in ((enc, "parent"), (child, "child")):
        comp = g_.fused_relations_complement("disease")
        assert comp is not None and comp[1] == cans, who
        ccsr, _, i0, blockmat = comp
        full = g_.fused_relations("disease")[0]
        ss = torch.rand(full.n_src)
        xa, xb = xf.clone().requires_grad_(True), xf.clone().requires_grad_(True)
        s_row = blockmat @ (ss.view(-1, len(cans))[:, i0:i0 + 1] * xa.view(-1, len(cans), 8)[:, i0])
        y_c = ops.spmm_csr(ccsr, torch.cat([xa, s_row]), torch.cat([ss, torch.ones(blockmat.shape[0])]))
        y_f = ops.spmm_csr(full, xb, ss)
        assert ccsr.n_src == full.n_src + blockmat.shape[0] == full.n_src + 1 and torch.allclose(y_c, y_f, atol=1e-5), who
        w = torch.randn(full.n_dst, 8)
        y_c.backward(w)
        y_f.backward(w)
        assert torch.allclose(xa.grad, xb.grad, atol=1e-5), who  # the virtual source's gradient flows back through colsum
    assert child.fused_relations_complement("disease")[0].nnz > enc.fused_relations_complement("disease")[0].nnz  # + relation i0
    assert child.fused_relations_complement("disease")[0].indptr is enc._complement_struct("disease", True)[0].indptr
    # dropout of a dropout falls back to materialised lists
    grand = G.random_edge_dropout(child, 0.5)
    assert not isinstance(grand["0"], G.DroppedRelation) and grand["0"].number_of_edges() == max(1, int(child["0"].number_of_edges() * 0.5))
    # sparse adjacency
    f = C.load("fgcn_both")
    nd = int(f["n_drug"])
    adj = C.sparse(f, "adj_d", nd, nd, CPU)
    dropped = G.random_edge_dropout_sparse(adj, 0.25)
    x = torch.randn(nd, 6)
    view = L.adjacency_csr(dropped)
    assert view.indptr is L.adjacency_csr(adj).indptr
    assert torch.allclose(view.spmm(x), torch.spmm(dropped, x), atol=1e-6)
    assert torch.allclose(view.spmm_t(x), torch.spmm(dropped.t(), x), atol=1e-6)
    # the view form (no sparse tensor is built): weighted adjacency x on-the-fly subset
    v2 = G.random_edge_dropout_sparse(adj, 0.25, as_view=True, selection=selection)
    base = L.adjacency_csr(adj)
    keep = v2.keep_mask() if selection == "select" else (v2._coo_vals != 0).float()
    assert int(keep.sum()) == max(1, int(base.nnz * 0.75))
    ref = ops.CSRGraph(base._S.dst, base._S.src, nd, nd, vals=base._coo_vals * keep)
    assert torch.allclose(v2.spmm(x), ref.spmm(x), atol=1e-6) and torch.allclose(v2.spmm_t(x), ref.spmm_t(x), atol=1e-6)
    # a dropout of the dropped view composes as the reference does (augmentation.py:113-124 on its own output): an
    # exact max(1, int(E' * (1 - p))) of the E' survivors, all of them survivors of the first dropout
    E1 = int(keep.sum())
    v3 = G.random_edge_dropout_sparse(v2, 0.5, as_view=True, selection=selection)
    keep3 = v3.survivors()
    assert int(keep3.sum()) == max(1, int(E1 * 0.5)) and bool((keep3 <= keep).all())
    ref3 = ops.CSRGraph(base._S.dst, base._S.src, nd, nd, vals=base._coo_vals * keep3)
    assert torch.allclose(v3.spmm(x), ref3.spmm(x), atol=1e-6) and torch.allclose(v3.spmm_t(x), ref3.spmm_t(x), atol=1e-6)
    v4 = G.random_edge_dropout_sparse_views([v2], 0.5)[0]
    assert int(v4.survivors().sum()) == max(1, int(E1 * 0.5))
    # ... and through sparse tensors: the second tensor's entries index the ROOT's CSR
    d2 = G.random_edge_dropout_sparse(dropped, 0.5)
    assert d2._values().shape[0] == max(1, int(dropped._values().shape[0] * 0.5))
    view2 = L.adjacency_csr(d2)
    assert view2.indptr is base.indptr and torch.allclose(view2.spmm(x), torch.spmm(d2, x), atol=1e-6)
    # CSRGraph.dropped on a dropped view: the descriptions are ANDed (intersection), never silently ignored
    if selection == "select":
        d_a = ops.random_subset_select(base.nnz, base.nnz // 2, 11, CPU)
        d_b = ops.random_subset_select(base.nnz, base.nnz // 2, 12, CPU)
        both = base.dropped(d_a).dropped(d_b)
        m = ops.keep_mask(d_a, base.nnz) * ops.keep_mask(d_b, base.nnz)
        assert torch.equal(both.keep_mask(), m) and 0 < int(m.sum()) < base.nnz // 2
        ref_ab = ops.CSRGraph(base._S.dst, base._S.src, nd, nd, vals=base._coo_vals * m)
        assert torch.allclose(both.spmm(x), ref_ab.spmm(x), atol=1e-6)
        # masked(m).dropped(d) must carry the values masked() multiplied its mask into — a dropout
        # of that view (random_edge_dropout_sparse: survivors() is not None -> undropped().masked(...)) starts from the
        # ORIGINAL weights, never from unit weights
        m0 = (torch.rand(base.nnz) < 0.8).float()
        md = base.masked(m0).dropped(d_a)
        assert md._vals_before_mask is base._coo_vals and torch.equal(md.undropped()._coo_vals, base._coo_vals)
        again = G.random_edge_dropout_sparse(md, 0.5, as_view=True)
        alive = again.survivors()
        assert bool((alive <= m0 * ops.keep_mask(d_a, base.nnz)).all()) and int(alive.sum()) == max(1, int(int((m0 * ops.keep_mask(d_a, base.nnz)).sum()) * 0.5))
        ref_again = ops.CSRGraph(base._S.dst, base._S.src, nd, nd, vals=base._coo_vals * alive)
        assert torch.allclose(again.spmm(x), ref_again.spmm(x), atol=1e-6)  # the adjacency's own weights, not ones


def test_complement_form_on_a_block_diagonal_union():
    """BASELINE config 3's shape in miniature: two datasets side by side, every train pair of each a cell of its own
    block, nothing across.  The label-0 relation is near-complete PER BLOCK: blocks are found as connected
    components, each gets its own column sum, and the complement aggregate equals the plain fused one (values
    and gradients).  A graph that is not dense per block is left alone."""
    from dream_gnn_amd import graph as G, ops

    rng = np.random.default_rng(4)
    blocks = [(9, 7), (6, 5)]
    drug, dis, lab = [], [], []
    d0 = s0 = 0
    for nd, ns in blocks:
        cells = [(d, s_) for d in range(nd) for s_ in range(ns)]
        keep = rng.permutation(len(cells))[: int(0.9 * len(cells))]
        for k in keep:
            drug.append(cells[k][0] + d0)
            dis.append(cells[k][1] + s0)
            lab.append(1.0 if rng.random() < 0.08 else 0.0)
        d0, s0 = d0 + nd, s0 + ns
    drug.append(d0)  # plus an isolated pair of nodes joined by a single label-1 edge: no block of relation 0
    dis.append(s0)
    lab.append(1.0)
    enc = G.build_enc_graph(torch.tensor(drug), torch.tensor(dis), torch.tensor(lab), d0 + 1, s0 + 1, device=CPU).int()
    present = torch.zeros(s0 + 1, d0 + 1, dtype=torch.bool)
    sel = torch.tensor(lab) == 0
    present[torch.tensor(dis)[sel], torch.tensor(drug)[sel]] = True
    blk_dst, blk_src, B = G._bipartite_blocks(present)
    assert B == 2 and blk_src[:9].unique().numel() == 1 and blk_src[9:15].unique().numel() == 1 and blk_src[15] == -1
    assert blk_dst[:7].unique().numel() == 1 and blk_dst[7:12].unique().numel() == 1 and blk_dst[12] == -1
    assert blk_src[0] != blk_src[9] and blk_dst[0] == blk_src[0] and blk_dst[7] == blk_src[9]
    assert G._bipartite_blocks(torch.eye(20, dtype=torch.bool)) is None  # 20 components: not a union of dense blocks
    for nt in ("disease", "drug"):
        full, cans = enc.fused_relations(nt)
        ccsr, cans2, i0, blockmat = enc.fused_relations_complement(nt)
        R = len(cans)
        n_src = full.n_src // R
        assert cans2 == cans and blockmat.shape == (2, n_src) and ccsr.nnz < full.nnz and ccsr.n_src == full.n_src + 2
        xa, xb = torch.randn(full.n_src, 6).requires_grad_(True), None
        xb = xa.detach().clone().requires_grad_(True)
        ss, ds = torch.rand(full.n_src) + 0.5, torch.rand(full.n_dst) + 0.5
        s_rows = blockmat @ (ss.view(-1, R)[:, i0:i0 + 1] * xa.view(n_src, R, 6)[:, i0])
        y_c = ops.spmm_csr(ccsr, torch.cat([xa, s_rows]), torch.cat([ss, torch.ones(2)]), ds)
        y_f = ops.spmm_csr(full, xb, ss, ds)
        assert torch.allclose(y_c, y_f, atol=1e-5)
        w = torch.randn_like(y_f)
        y_c.backward(w)
        y_f.backward(w)
        assert torch.allclose(xa.grad, xb.grad, atol=1e-5)
    cells = torch.from_numpy(rng.choice(30 * 30, 60, replace=False))  # 7 % of the cells, one big sparse component
    sparse = G.build_enc_graph(cells // 30, cells % 30, torch.zeros(60), 30, 30, device=CPU).int()
    assert sparse.fused_relations_complement("disease") is None and sparse.fused_relations_complement("drug") is None


def test_random_subset_selection_is_exact_and_uniformish(oracle):
    """oracle.random_subset_mask (restating dgmi_random_subset_mask_f32): exact size, deterministic
    in the seed, different seeds differ, inclusion frequency ~ keep / E."""
    E, keep = 5000, 1234
    m1, m2 = oracle.random_subset_mask(E, keep, 42), oracle.random_subset_mask(E, keep, 42)
    assert m1.sum() == keep and np.array_equal(m1, m2) and set(np.unique(m1)) <= {0.0, 1.0}
    assert not np.array_equal(m1, oracle.random_subset_mask(E, keep, 43))
    assert oracle.random_subset_mask(E, 0, 1).sum() == 0 and oracle.random_subset_mask(E, E, 1).sum() == E
    freq = sum(oracle.random_subset_mask(400, 100, s) for s in range(300)) / 300.0
    assert abs(freq.mean() - 0.25) < 1e-9 and freq.min() > 0.12 and freq.max() < 0.40


def test_decoder_edge_linear_split_k_gradients_equal_the_plain_linear():
    """model._EdgeLinear: the decoder's Linear layers over the edge list with the weight gradient summed chunk by
    chunk (K = number of train pairs) — same forward, same gradients as nn.Linear up to fp32 summation order; the
    number of rows is deliberately not a multiple of the chunk."""
    from dream_gnn_amd import model as M

    torch.manual_seed(3)
    lin = torch.nn.Linear(24, 10)
    x = torch.randn(3 * M._EdgeLinear.CHUNK + 517, 24, requires_grad=True)
    w = torch.randn(x.shape[0], 10)
    y_ref = lin(x)
    (y_ref * w).sum().backward()
    ref = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    y = M._EdgeLinear.apply(x, lin.weight, lin.bias)
    (y * w).sum().backward()
    assert torch.equal(y, y_ref)
    assert torch.equal(x.grad, ref[0])
    for got, want in ((lin.weight.grad, ref[1]), (lin.bias.grad, ref[2])):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # no bias (lin3-style single output), and the dispatcher's size / device gate
    lin1 = torch.nn.Linear(24, 1, bias=False)
    x2 = x.detach().clone().requires_grad_(True)
    M._EdgeLinear.apply(x2, lin1.weight, None).sum().backward()
    want = x.detach().sum(0, keepdim=True)
    assert float((lin1.weight.grad - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert M._edge_linear(lin, x[:8]).shape == (8, 10)  # CPU / short inputs take nn.Linear itself
