#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE in place (build container only).

    python tests/golden/gen_golden.py            # needs /root/reference; never runs on the GPU box

Nothing from /root/reference is copied: the script imports the reference's modules from where
they lie and stores only inputs / parameters / outputs as arrays.

Provenance of each file (what produced the expected values):
  csr_*.npz        numpy ``argsort(kind='stable')``                       (independent of oracle/)
  spmm_*.npz       weighted: the installed ``torch.spmm`` + its autograd  (reference layers.py:312)
                   unweighted: ``y_copy_u`` = ``dgl_standin`` update_all (torch index_add) — NOT DGL,
                   see below — and beside it ``y_unit_spmm`` = the installed ``torch.spmm`` on the same
                   edges with unit values (independent third-party code for the same sum)
  simgraph_*.npz   reference ``DrugDataLoader._create_similarity_graph`` + ``utils.normalize`` +
                   ``utils.sparse_mx_to_torch_sparse_tensor``, ``augmentation.random_edge_dropout_sparse``
  encgraph_*.npz   reference ``DrugDataLoader._generate_enc_graph`` (ci/cj)  [graph ctor: stand-in]
  gcmc_conv_*.npz  reference ``layers.GCMCGraphConv``                        [primitive: stand-in]
  gcmc_layer_*.npz reference ``layers.GCMCLayer``                            [primitive: stand-in]
  fgcn_*.npz       reference ``layers.GraphConvolution / GCN / FGCN``        (torch.spmm, no stand-in)
  net_*.npz        reference ``model.Net`` forward + loss + grads            [primitive: stand-in]
  metrics.npz      reference ``evaluation.evaluate`` metric lines (sklearn)
  common_loss.npz  reference ``utils.common_loss``
DGL itself is absent (not installable here): files marked [stand-in] pin the reference's own
Python around the primitive, not DGL's kernel ("parity unpinned" for that kernel).
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DREAMGNN_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import dgl_standin  # noqa: E402

dgl = dgl_standin.install()

import utils as ref_utils  # noqa: E402  (imports as-is)
import evaluation as ref_eval  # noqa: E402  (imports as-is)
import layers as ref_layers  # noqa: E402  (needs the dgl stand-in)
import augmentation as ref_aug  # noqa: E402
import data_loader as ref_dl  # noqa: E402
import model as ref_model  # noqa: E402

th.set_printoptions(profile="default")
th.set_num_threads(1)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, th.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------------------
def gen_csr():
    rng = np.random.default_rng(100)
    cases = {
        "random": (rng.integers(0, 37, 500), rng.integers(0, 53, 500), 37),
        "dups": (np.array([3, 3, 3, 0, 0, 5, 3]), np.array([1, 1, 1, 2, 2, 0, 1]), 7),
        "empty_rows": (np.array([6, 6, 2, 2, 2]), np.array([0, 1, 2, 3, 4]), 9),
        "single_row": (np.zeros(40, np.int64), rng.integers(0, 5, 40), 1),
        "E0": (np.zeros(0, np.int64), np.zeros(0, np.int64), 4),
    }
    for name, (row, col, n) in cases.items():
        row = row.astype(np.int32)
        col = col.astype(np.int32)
        order = np.argsort(row, kind="stable").astype(np.int32)
        indptr = np.zeros(n + 1, np.int32)
        np.add.at(indptr, row + 1, 1)
        indptr = np.cumsum(indptr).astype(np.int32)
        save("csr_" + name, row=row, col=col, n_rows=n, indptr=indptr, indices=col[order], eid=order)


def gen_spmm():
    for F in (1, 3, 4, 127, 128, 341, 768):
        rng = np.random.default_rng(200 + F)
        n_dst, n_src, E = (24, 20, 160) if F > 200 else (48, 40, 500)
        dst = rng.integers(0, n_dst, E)
        src = rng.integers(0, n_src, E)
        dst[dst == 3] = 2  # an empty row
        val = rng.standard_normal(E).astype(np.float32)
        X = rng.standard_normal((n_src, F)).astype(np.float32)
        dY = rng.standard_normal((n_dst, F)).astype(np.float32)
        # weighted: torch.spmm on an UNCOALESCED COO in this (random) order — layers.py:312
        x = th.from_numpy(X).requires_grad_(True)
        adj = th.sparse_coo_tensor(th.from_numpy(np.vstack([dst, src])), th.from_numpy(val), (n_dst, n_src))
        y = th.spmm(adj, x)
        y.backward(th.from_numpy(dY))
        # unweighted: stand-in update_all(copy_u, sum) (NOT DGL)
        g = dgl.heterograph({("u", "e", "v"): (th.from_numpy(src), th.from_numpy(dst))},
                            num_nodes_dict={"u": n_src, "v": n_dst})
        x2 = th.from_numpy(X).requires_grad_(True)
        g.srcdata["h"] = x2
        g.update_all(dgl.function.copy_u("h", "m"), dgl.function.sum("m", "h"))
        y2 = g.dstdata["h"]
        y2.backward(th.from_numpy(dY))
        # the same unweighted sum through ATen: torch.spmm with unit values on the same COO
        x3 = th.from_numpy(X).requires_grad_(True)
        unit = th.sparse_coo_tensor(th.from_numpy(np.vstack([dst, src])), th.ones(E), (n_dst, n_src))
        y3 = th.spmm(unit, x3)
        y3.backward(th.from_numpy(dY))
        save("spmm_F%d" % F, dst=dst.astype(np.int32), src=src.astype(np.int32), val=val, X=X, dY=dY,
             n_dst=n_dst, n_src=n_src, y_weighted=y, dx_weighted=x.grad, y_copy_u=y2, dx_copy_u=x2.grad,
             y_unit_spmm=y3, dx_unit_spmm=x3.grad)


def _fake_loader(symm=True, n_drug=0, n_dis=0):
    return types.SimpleNamespace(_symm=symm, _num_drug=n_drug, _num_disease=n_dis)


def gen_simgraph():
    rng = np.random.default_rng(300)
    for name, n, k in (("n30_k4", 30, 4), ("n12_k20", 12, 20)):
        A = rng.random((n, n))
        sim = (A + A.T) / 2
        np.fill_diagonal(sim, 1.0)
        adj = ref_dl.DrugDataLoader._create_similarity_graph(_fake_loader(True), sim, k)
        th.manual_seed(1234)
        dropped = ref_aug.GraphAugmentation.random_edge_dropout_sparse(adj, 0.1)
        # utils.normalize on its own (zero row -> r_inv 0, utils.py:15)
        M = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [0.0, 0.0, 0.0], [4.0, 0.0, 4.0]]))
        save("simgraph_" + name, sim=sim, k=k, n=n,
             row=adj._indices()[0], col=adj._indices()[1], val=adj._values(),
             drop_row=dropped._indices()[0], drop_col=dropped._indices()[1], drop_val=dropped._values(),
             drop_rate=0.1, norm_in=M.toarray(), norm_out=ref_utils.normalize(M).toarray())


def _enc_inputs(rng, n_drug, n_dis, frac_pos=0.1, frac_used=0.9, isolate=True):
    """All-pairs style encoder input (data_loader.py:146-170): positives and negatives."""
    pairs = np.array([(d, s) for d in range(n_drug) for s in range(n_dis)])
    keep = rng.random(len(pairs)) < frac_used
    if isolate:
        keep &= pairs[:, 0] != n_drug - 1  # drug n_drug-1 is isolated -> ci = cj = 0
    pairs = pairs[keep]
    vals = (rng.random(len(pairs)) < frac_pos).astype(np.float32)
    vals[0], vals[1] = 0.0, 1.0
    return (pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)), vals


def gen_encgraph():
    rng = np.random.default_rng(400)
    n_drug, n_dis = 13, 9
    pairs, vals = _enc_inputs(rng, n_drug, n_dis)
    for symm in (True, False):
        g = ref_dl.DrugDataLoader._generate_enc_graph(_fake_loader(symm, n_drug, n_dis), pairs, vals,
                                                      add_support=True)
        arrays = dict(drug_ids=pairs[0], dis_ids=pairs[1], values=vals, n_drug=n_drug, n_dis=n_dis, symm=symm)
        for nt in ("drug", "disease"):
            for key in ("ci", "cj"):
                arrays["%s_%s" % (nt, key)] = g.nodes[nt].data[key]
        for can in g.canonical_etypes:
            s, d = g.edges(etype=can)
            arrays["src_" + can[1]] = s
            arrays["dst_" + can[1]] = d
        # edge dropout bookkeeping (augmentation.py:13-89)
        th.manual_seed(77)
        gd = ref_aug.GraphAugmentation.random_edge_dropout(g, 0.1)
        for can in gd.canonical_etypes:
            arrays["drop_n_" + can[1]] = gd.number_of_edges(can)
        arrays["drop_drug_ci"] = gd.nodes["drug"].data["ci"]
        save("encgraph_symm%d" % int(symm), **arrays)
    return


def _grads(module):
    return {"grad_" + k: p.grad for k, p in module.named_parameters() if p.grad is not None}


def _build_enc(rng, n_drug, n_dis):
    pairs, vals = _enc_inputs(rng, n_drug, n_dis)
    g = ref_dl.DrugDataLoader._generate_enc_graph(_fake_loader(True, n_drug, n_dis), pairs, vals, add_support=True)
    return pairs, vals, g.int()


def gen_gcmc_conv():
    rng = np.random.default_rng(500)
    n_drug, n_dis, fin, fout = 17, 11, 12, 8
    pairs, vals, g = _build_enc(rng, n_drug, n_dis)
    feat = rng.standard_normal((n_drug, fin)).astype(np.float32)
    dY = rng.standard_normal((n_dis, fout)).astype(np.float32)
    for mode in ("eval", "train"):
        th.manual_seed(9)
        conv = ref_layers.GCMCGraphConv(fin, fout, weight=True, dropout_rate=0.3)
        conv.train(mode == "train")
        rel = g["drug", "0", "disease"]
        x = th.from_numpy(feat).requires_grad_(True)
        th.manual_seed(4242)
        y = conv(rel, (x, None))
        y.backward(th.from_numpy(dY))
        # the multiplicative mask nn.Dropout drew on cj (same seed, same shape)
        th.manual_seed(4242)
        mask = th.nn.Dropout(0.3).train(mode == "train")(th.ones(n_drug, 1))
        s, d = rel.edges()
        save("gcmc_conv_" + mode, src=s, dst=d, n_src=n_drug, n_dst=n_dis, feat=feat, dY=dY,
             weight=conv.weight, cj=rel.srcdata["cj"], ci=rel.dstdata["ci"], cj_mask=mask,
             y=y, dfeat=x.grad, dweight=conv.weight.grad)


def gen_gcmc_layer():
    rng = np.random.default_rng(600)
    n_drug, n_dis = 17, 11
    pairs, vals, g = _build_enc(rng, n_drug, n_dis)
    cfgs = {
        # name: (user_in, movie_in, msg_units, out_units, ini, share, agg_act)
        "shared_ini": (12, 12, 24, 6, True, True, "leaky"),       # msg 24//3 = 8, W from att@basis
        "shared_noini": (12, 12, 16, 6, False, True, "leaky"),
        "unshared": (12, 10, 24, 6, True, False, "relu"),           # own weights per etype, ifc != ufc
        "shareflag_dimdiff": (12, 10, 24, 6, True, True, None),     # share flag but dims differ -> own weights
        # dropout on: pins the ORDER of the RNG draws (one (N_src,1) draw per relation in canonical
        # etype order, then the two layer-level draws) — CPU RNG stream, relative to the stand-in's
        # (sorted, DGL-style) relation order
        "shared_dropout": (12, 12, 16, 6, False, True, "leaky"),
    }
    for name, (uin, min_, msg, out, ini, share, act) in cfgs.items():
        th.manual_seed(11)
        p_drop = 0.3 if name == "shared_dropout" else 0.0
        layer = ref_layers.GCMCLayer([0, 1], uin, min_, msg, out, dropout_rate=p_drop, agg="sum",
                                     agg_act=ref_utils.get_activation(act), ini=ini,
                                     share_user_item_param=share)
        layer.train()
        drug = th.from_numpy(rng.standard_normal((n_drug, uin)).astype(np.float32)).requires_grad_(True)
        dis = th.from_numpy(rng.standard_normal((n_dis, min_)).astype(np.float32)).requires_grad_(True)
        d_drug = th.from_numpy(rng.standard_normal((n_drug, out)).astype(np.float32))
        d_dis = th.from_numpy(rng.standard_normal((n_dis, out)).astype(np.float32))
        th.manual_seed(31337)  # the forward's dropout draws start from here
        o_drug, o_dis = layer(g, drug, dis)
        ((o_drug * d_drug).sum() + (o_dis * d_dis).sum()).backward()
        arrays = dict(drug_ids=pairs[0], dis_ids=pairs[1], values=vals, n_drug=n_drug, n_dis=n_dis,
                      cfg=np.array([uin, min_, msg, out, int(ini), int(share)]), act=str(act),
                      drug=drug, dis=dis, d_drug=d_drug, d_dis=d_dis, o_drug=o_drug, o_dis=o_dis,
                      g_drug=drug.grad, g_dis=dis.grad)
        for k, v in layer.state_dict().items():
            arrays["sd_" + k] = v
        arrays.update(_grads(layer))
        save("gcmc_layer_" + name, **arrays)


def _sim_adj(rng, n, k):
    A = rng.random((n, n))
    sim = (A + A.T) / 2
    np.fill_diagonal(sim, 1.0)
    return ref_dl.DrugDataLoader._create_similarity_graph(_fake_loader(True), sim, k)


def gen_fgcn():
    rng = np.random.default_rng(700)
    n_drug, n_dis, nhid1, nhid2 = 21, 15, 16, 8
    adj_d, adj_s = _sim_adj(rng, n_drug, 4), _sim_adj(rng, n_dis, 4)
    th.manual_seed(5)
    fadj_d = ref_aug.GraphAugmentation.random_edge_dropout_sparse(_sim_adj(rng, n_drug, 4), 0.1)  # shuffled, uncoalesced
    fadj_s = ref_aug.GraphAugmentation.random_edge_dropout_sparse(_sim_adj(rng, n_dis, 4), 0.1)
    xd = rng.standard_normal((n_drug, n_drug)).astype(np.float32)
    xs = rng.standard_normal((n_dis, n_dis)).astype(np.float32)
    for name, with_feat in (("both", True), ("simonly", False)):
        th.manual_seed(6)
        net = ref_layers.FGCN(n_drug, n_dis, nhid1, nhid2, dropout=0.0)
        net.train()
        a = th.from_numpy(xd).requires_grad_(True)
        b = th.from_numpy(xs).requires_grad_(True)
        outs = net(adj_d, a, adj_s, b, fadj_d if with_feat else None, fadj_s if with_feat else None)
        w1 = th.from_numpy(rng.standard_normal(outs[0].shape).astype(np.float32))
        w2 = th.from_numpy(rng.standard_normal(outs[1].shape).astype(np.float32))
        ((outs[0] * w1).sum() + (outs[1] * w2).sum()).backward()
        arrays = dict(n_drug=n_drug, n_dis=n_dis, nhid1=nhid1, nhid2=nhid2, xd=xd, xs=xs, w1=w1, w2=w2,
                      g_xd=a.grad, g_xs=b.grad)
        for nm, adj in (("adj_d", adj_d), ("adj_s", adj_s), ("fadj_d", fadj_d), ("fadj_s", fadj_s)):
            arrays[nm + "_row"], arrays[nm + "_col"], arrays[nm + "_val"] = adj._indices()[0], adj._indices()[1], adj._values()
        for i, o in enumerate(outs):
            if o is not None:
                arrays["out%d" % i] = o
        for k, v in net.state_dict().items():
            arrays["sd_" + k] = v
        arrays.update(_grads(net))
        save("fgcn_" + name, **arrays)
    # a single GraphConvolution without bias
    th.manual_seed(8)
    gc = ref_layers.GraphConvolution(n_drug, 5, bias=False)
    y = gc(th.from_numpy(xd), fadj_d)
    save("graphconv_nobias", x=xd, row=fadj_d._indices()[0], col=fadj_d._indices()[1], val=fadj_d._values(),
         n=n_drug, weight=gc.weight, y=y)


def gen_net():
    """lrssl-like miniature: all-pairs encoder graph, k=4 sim graphs, 3 GCMC layers + FGCN."""
    rng = np.random.default_rng(800)
    n_drug, n_dis, emb = 23, 19, 24
    pairs, vals, enc = _build_enc(rng, n_drug, n_dis)
    dec = ref_dl.DrugDataLoader._generate_dec_graph(_fake_loader(True, n_drug, n_dis), pairs).int()
    drug_graph, dis_graph = _sim_adj(rng, n_drug, 4), _sim_adj(rng, n_dis, 4)
    drug_fg, dis_fg = _sim_adj(rng, n_drug, 4), _sim_adj(rng, n_dis, 4)
    drug_feat = th.nn.functional.normalize(th.from_numpy(rng.standard_normal((n_drug, emb)).astype(np.float32)))
    dis_feat = th.nn.functional.normalize(th.from_numpy(rng.standard_normal((n_dis, emb)).astype(np.float32)))
    drug_sim = th.from_numpy(rng.random((n_drug, n_drug)).astype(np.float32))
    dis_sim = th.from_numpy(rng.random((n_dis, n_dis)).astype(np.float32))
    args = types.SimpleNamespace(rating_vals=[0, 1], src_in_units=emb, dst_in_units=emb, gcn_agg_units=48,
                                 gcn_out_units=8, dropout=0.0, gcn_agg_accum="sum", model_activation="leaky",
                                 share_param=True, device=None, layers=3, fdim_drug=n_drug, fdim_disease=n_dis,
                                 nhid1=16, nhid2=8, attention_dropout=0.0, beta=0.1)
    th.manual_seed(21)
    net = ref_model.Net(args)
    net.train()
    pred, drug_out, drug_sim_out, dis_out, dis_sim_out = net(enc, dec, drug_graph, drug_sim, drug_feat, dis_graph,
                                                             dis_sim, dis_feat, drug_fg, dis_fg)
    labels = th.from_numpy(vals)
    loss = th.nn.BCEWithLogitsLoss()(pred.squeeze(-1), labels) + args.beta * (
        ref_utils.common_loss(drug_out, drug_sim_out) + ref_utils.common_loss(dis_out, dis_sim_out))
    loss.backward()
    arrays = dict(drug_ids=pairs[0], dis_ids=pairs[1], values=vals, n_drug=n_drug, n_dis=n_dis, emb=emb,
                  drug_feat=drug_feat, dis_feat=dis_feat, drug_sim=drug_sim, dis_sim=dis_sim,
                  pred=pred, drug_out=drug_out, drug_sim_out=drug_sim_out, dis_out=dis_out,
                  dis_sim_out=dis_sim_out, loss=loss)
    s, d = dec.edges()
    arrays["dec_src"], arrays["dec_dst"] = s, d
    for nm, adj in (("drug_graph", drug_graph), ("dis_graph", dis_graph), ("drug_fg", drug_fg), ("dis_fg", dis_fg)):
        arrays[nm + "_row"], arrays[nm + "_col"], arrays[nm + "_val"] = adj._indices()[0], adj._indices()[1], adj._values()
    for k, v in net.state_dict().items():
        arrays["sd_" + k] = v
    arrays.update(_grads(net))
    save("net_mini", **arrays)

    # metric lines of evaluation.py:55-65 through the reference's own evaluate()
    class _FakeModel:
        def eval(self):
            return self

        def __call__(self, *a, **k):
            return self.scores, None, None, None, None

    class _G:
        def int(self):
            return self

        def to(self, _):
            return self

    m = _FakeModel()
    rs = np.random.default_rng(900)
    ys, yt, au, ap = [], [], [], []
    for n, pos in ((50, 0.3), (400, 0.05), (64, 0.5)):
        y_true = (rs.random(n) < pos).astype(np.float32)
        y_true[:2] = [0, 1]
        score = (rs.standard_normal(n) + 1.5 * y_true).astype(np.float32)
        score[:8] = np.round(score[:8])  # ties
        m.scores = th.from_numpy(score).view(-1, 1)
        dummy = th.zeros(1)
        auc, aupr = ref_eval.evaluate(types.SimpleNamespace(device="cpu"), m,
                                      {"test": [_G(), _G(), th.from_numpy(y_true)]},
                                      dummy, None, None, dummy, None, None)
        ys.append(score); yt.append(y_true); au.append(auc); ap.append(aupr)
    save("metrics", **{"score%d" % i: s for i, s in enumerate(ys)}, **{"true%d" % i: t for i, t in enumerate(yt)},
         auroc=np.array(au), aupr=np.array(ap))

    e1 = th.from_numpy(rs.standard_normal((9, 5)).astype(np.float32))
    e2 = th.from_numpy(rs.standard_normal((9, 5)).astype(np.float32))
    save("common_loss", e1=e1, e2=e2, loss=ref_utils.common_loss(e1, e2))


if __name__ == "__main__":
    gen_csr()
    gen_spmm()
    gen_simgraph()
    gen_encgraph()
    gen_gcmc_conv()
    gen_gcmc_layer()
    gen_fgcn()
    gen_net()
