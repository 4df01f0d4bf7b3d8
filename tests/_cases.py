"""Module-level parity cases shared by the CPU host-logic tests (oracle backend patched in)
and the GPU tests (HIP path).  Expected values come from tests/golden (reference run in place).

Tolerance: 1e-5 relative to the tensor's max magnitude for embeddings (north_star), 1e-4 for
parameter gradients that pass through several fp32 GEMMs.
"""
import os
import types

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def close(a, b, rtol=1e-5, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if b.size else 0.0
    assert err <= rtol * scale + 1e-7, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def sparse(g, prefix, n_rows, n_cols, dev):
    idx = torch.from_numpy(np.vstack([g[prefix + "_row"], g[prefix + "_col"]]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(g[prefix + "_val"]), (n_rows, n_cols)).to(dev)


def load_sd(module, g, dev):
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    missing = module.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return module.to(dev)


def check_grads(module, g, rtol=1e-4):
    n = 0
    for k, p in module.named_parameters():
        key = "grad_" + k
        if key in g.files:
            assert p.grad is not None, k
            close(p.grad, g[key], rtol, key)
            n += 1
    assert n == sum(1 for k in g.files if k.startswith("grad_")), "some reference grads have no counterpart"


# ---------------------------------------------------------------------------------------------
def case_csr(dev, name):
    from dream_gnn_amd import ops

    g = load("csr_" + name)
    indptr, indices, eid = ops.csr_from_coo(T(g["row"], dev), T(g["col"], dev), int(g["n_rows"]))
    assert np.array_equal(indptr.cpu().numpy(), g["indptr"])  # bit-exact
    assert np.array_equal(indices.cpu().numpy(), g["indices"])
    assert np.array_equal(eid.cpu().numpy(), g["eid"])


def case_spmm(dev, F):
    from dream_gnn_amd import ops

    g = load("spmm_F%d" % F)
    n_dst, n_src = int(g["n_dst"]), int(g["n_src"])
    for weighted in (True, False):
        graph = ops.CSRGraph(T(g["dst"], dev), T(g["src"], dev), n_dst, n_src,
                             vals=T(g["val"], dev) if weighted else None)
        x = T(g["X"], dev).requires_grad_(True)
        y = ops.spmm_csr(graph, x)
        y.backward(T(g["dY"], dev))
        tag = "weighted" if weighted else "copy_u"
        close(y, g["y_" + tag], 1e-5, "y_" + tag)
        close(x.grad, g["dx_" + tag], 1e-5, "dx_" + tag)
        if not weighted:  # the same sum through ATen (torch.spmm on unit values): third-party pin
            close(y, g["y_unit_spmm"], 1e-5, "y_unit_spmm")
            close(x.grad, g["dx_unit_spmm"], 1e-5, "dx_unit_spmm")
        deg = np.bincount(g["dst"], minlength=n_dst)
        assert np.all(y.detach().cpu().numpy()[deg == 0] == 0)


def case_gcmc_conv(dev, mode):
    from dream_gnn_amd import graph as G, layers as L

    g = load("gcmc_conv_" + mode)
    n_src, n_dst = int(g["n_src"]), int(g["n_dst"])
    hg = G.HeteroGraph({("drug", "0", "disease"): (T(g["src"], dev), T(g["dst"], dev))},
                       {"drug": n_src, "disease": n_dst}).int()
    hg.nodes["drug"].data["cj"] = T(g["cj"], dev)
    hg.nodes["disease"].data["ci"] = T(g["ci"], dev)
    conv = L.GCMCGraphConv(g["feat"].shape[1], g["weight"].shape[1], weight=True, dropout_rate=0.3)
    conv.weight.data = torch.from_numpy(g["weight"])
    conv = conv.to(dev)
    conv.train(mode == "train")
    mask = T(g["cj_mask"], dev)

    class _FixedMask(torch.nn.Module):  # replays the mask nn.Dropout drew in the reference run
        def forward(self, x):
            return x * mask

    conv.dropout = _FixedMask()
    x = T(g["feat"], dev).requires_grad_(True)
    rel = hg["drug", "0", "disease"]
    y = conv(rel, (x, None))
    y.backward(T(g["dY"], dev))
    close(y, g["y"], 1e-5, "y")
    close(x.grad, g["dfeat"], 1e-5, "dfeat")
    close(conv.weight.grad, g["dweight"], 1e-5, "dweight")
    assert "h" not in rel.srcdata and "h" not in rel.dstdata  # local_scope semantics


def build_enc(g, dev):
    from dream_gnn_amd import graph as G

    return G.build_enc_graph(torch.from_numpy(g["drug_ids"]), torch.from_numpy(g["dis_ids"]),
                             torch.from_numpy(g["values"]), int(g["n_drug"]), int(g["n_dis"]),
                             symm=True, device=dev).int()


def case_gcmc_layer(dev, name, fuse=True, device_arg=None, dropout_rate=0.0, seed=None, complement=True):
    from dream_gnn_amd import layers as L

    g = load("gcmc_layer_" + name)
    uin, min_, msg, out, ini, share = [int(v) for v in g["cfg"]]
    act = str(g["act"])
    layer = L.GCMCLayer([0, 1], uin, min_, msg, out, dropout_rate=dropout_rate, agg="sum",
                        agg_act=L.get_activation(None if act == "None" else act), ini=bool(ini),
                        share_user_item_param=bool(share), device=device_arg)
    layer = load_sd(layer, g, dev)
    layer.train()
    layer.fuse_relations = fuse
    layer.complement_form = complement
    enc = build_enc(g, dev)
    drug = T(g["drug"], dev).requires_grad_(True)
    dis = T(g["dis"], dev).requires_grad_(True)
    if seed is not None:
        torch.manual_seed(seed)
    o_drug, o_dis = layer(enc, drug, dis)
    ((o_drug * T(g["d_drug"], dev)).sum() + (o_dis * T(g["d_dis"], dev)).sum()).backward()
    fused = enc.__dict__.get("_fused", {})
    assert (set(fused) == {"drug", "disease"} and all(v is not None for v in fused.values())) == fuse
    # the fixtures are reference-shaped (156 of 187 cells are train pairs, label 0 alone > 50 %): the fused path runs
    # in complement form (graph.fused_relations_complement) unless told not to
    comp = enc.__dict__.get("_fused_complement", {})
    assert (set(comp) == {"drug", "disease"} and all(v is not None for v in comp.values())) == (fuse and complement)
    close(o_drug, g["o_drug"], 1e-5, "o_drug")
    close(o_dis, g["o_dis"], 1e-5, "o_dis")
    close(drug.grad, g["g_drug"], 1e-4, "g_drug")
    close(dis.grad, g["g_dis"], 1e-4, "g_dis")
    check_grads(layer, g)


def case_fgcn(dev, name):
    from dream_gnn_amd import layers as L

    g = load("fgcn_" + name)
    nd, ns = int(g["n_drug"]), int(g["n_dis"])
    net = load_sd(L.FGCN(nd, ns, int(g["nhid1"]), int(g["nhid2"]), dropout=0.0), g, dev)
    net.train()
    a = T(g["xd"], dev).requires_grad_(True)
    b = T(g["xs"], dev).requires_grad_(True)
    with_feat = name == "both"
    outs = net(sparse(g, "adj_d", nd, nd, dev), a, sparse(g, "adj_s", ns, ns, dev), b,
               sparse(g, "fadj_d", nd, nd, dev) if with_feat else None,
               sparse(g, "fadj_s", ns, ns, dev) if with_feat else None)
    ((outs[0] * T(g["w1"], dev)).sum() + (outs[1] * T(g["w2"], dev)).sum()).backward()
    for i, o in enumerate(outs):
        if "out%d" % i in g.files:
            close(o, g["out%d" % i], 1e-5, "out%d" % i)
        else:
            assert o is None
    close(a.grad, g["g_xd"], 1e-4, "g_xd")
    close(b.grad, g["g_xs"], 1e-4, "g_xs")
    check_grads(net, g)


def case_graphconv_nobias(dev):
    from dream_gnn_amd import layers as L

    g = load("graphconv_nobias")
    n = int(g["n"])
    gc = L.GraphConvolution(n, g["weight"].shape[1], bias=False)
    gc.weight.data = torch.from_numpy(g["weight"])
    gc = gc.to(dev)
    idx = torch.from_numpy(np.vstack([g["row"], g["col"]]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g["val"]), (n, n)).to(dev)
    close(gc(T(g["x"], dev), adj), g["y"], 1e-5, "y")
    assert "bias" in dict(gc.named_parameters()) or gc.bias is None


def net_args(g):
    return types.SimpleNamespace(rating_vals=[0, 1], src_in_units=int(g["emb"]), dst_in_units=int(g["emb"]),
                                 gcn_agg_units=48, gcn_out_units=8, dropout=0.0, gcn_agg_accum="sum",
                                 model_activation="leaky", share_param=True, device=None, layers=3,
                                 fdim_drug=int(g["n_drug"]), fdim_disease=int(g["n_dis"]), nhid1=16, nhid2=8,
                                 attention_dropout=0.0, beta=0.1)


def case_net(dev, fuse_decoder=True):
    from dream_gnn_amd import graph as G, model as M

    g = load("net_mini")
    nd, ns = int(g["n_drug"]), int(g["n_dis"])
    args = net_args(g)
    net = load_sd(M.Net(args), g, dev)  # the reference's state_dict loads strict=True
    net.train()
    net.decoder.fuse_lin1 = fuse_decoder
    enc = build_enc(g, dev)
    dec = G.build_dec_graph(torch.from_numpy(g["dec_src"]), torch.from_numpy(g["dec_dst"]), nd, ns, device=dev).int()
    pred, drug_out, drug_sim_out, dis_out, dis_sim_out = net(
        enc, dec, sparse(g, "drug_graph", nd, nd, dev), T(g["drug_sim"], dev), T(g["drug_feat"], dev),
        sparse(g, "dis_graph", ns, ns, dev), T(g["dis_sim"], dev), T(g["dis_feat"], dev),
        sparse(g, "drug_fg", nd, nd, dev), sparse(g, "dis_fg", ns, ns, dev))
    loss = torch.nn.BCEWithLogitsLoss()(pred.squeeze(-1), T(g["values"], dev)) + args.beta * (
        M.common_loss(drug_out, drug_sim_out) + M.common_loss(dis_out, dis_sim_out))
    loss.backward()
    for nm, t in (("pred", pred), ("drug_out", drug_out), ("drug_sim_out", drug_sim_out),
                  ("dis_out", dis_out), ("dis_sim_out", dis_sim_out)):
        close(t, g[nm], 1e-5, nm)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    check_grads(net, g, rtol=2e-4)


def case_encgraph(dev, symm):
    from dream_gnn_amd import graph as G

    g = load("encgraph_symm%d" % symm)
    hg = G.build_enc_graph(torch.from_numpy(g["drug_ids"]), torch.from_numpy(g["dis_ids"]),
                           torch.from_numpy(g["values"]), int(g["n_drug"]), int(g["n_dis"]),
                           symm=bool(symm), device=dev)
    assert sorted(hg.etypes) == ["0", "1", "rev-0", "rev-1"]
    for nt, short in (("drug", "drug"), ("disease", "disease")):
        for key in ("ci", "cj"):
            ref = g["%s_%s" % (short, key)]
            got = hg.nodes[nt].data[key].cpu().numpy()
            assert got.shape == ref.shape and np.array_equal(got, ref), (nt, key)  # exact: 1/sqrt(int)
    for et in hg.etypes:
        s, d = hg.edges(etype=et)
        assert np.array_equal(s.cpu().numpy(), g["src_" + et]) and np.array_equal(d.cpu().numpy(), g["dst_" + et])
    # edge dropout keeps max(1, int(E*(1-p))) edges per relation and copies (stale) norms
    dropped = G.random_edge_dropout(hg, 0.1)
    for et in hg.etypes:
        assert dropped.number_of_edges(et) == int(g["drop_n_" + et])
    assert np.array_equal(dropped.nodes["drug"].data["ci"].cpu().numpy(), g["drop_drug_ci"])
    assert dropped.nodes["drug"].data["ci"].data_ptr() != hg.nodes["drug"].data["ci"].data_ptr()


def case_similarity_graph(dev, name):
    """(f4) device kNN-graph builder vs the reference's _create_similarity_graph output."""
    from dream_gnn_amd import graph as G

    g = load("simgraph_" + name)
    adj = G.similarity_graph(T(g["sim"], dev), int(g["k"])).coalesce()
    idx = adj.indices().cpu().numpy()
    assert np.array_equal(idx[0], g["row"]) and np.array_equal(idx[1], g["col"])  # same edges, same order
    assert np.array_equal(adj.values().cpu().numpy(), g["val"])  # float32(count / rowsum) exactly
    # the cosine-feature variant agrees with the dense-similarity path on the same embeddings
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((int(g["n"]), 16)).astype(np.float32)).to(dev)
    xn = X / X.norm(dim=1, keepdim=True)
    a = G.feature_similarity_graph(X, 3, block_rows=7).coalesce()
    b = G.similarity_graph(xn @ xn.t(), 3).coalesce()
    assert torch.equal(a.indices(), b.indices()) and torch.equal(a.values(), b.values())
