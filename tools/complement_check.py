"""(f3) Complement form of the near-complete relation vs the plain fused CSR at the lrssl shape, one GCMCLayer:
  * parameter-gradient error of both against an f64 dense-matrix evaluation of the same layer (CPU),
  * us per forward / forward+backward, eval and train (10 % edge dropout on the fly)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dream_gnn_amd import graph as G, layers as L, synth

dev = torch.device("cuda:0")
drug, dis, labels, nd, ns = synth.dataset_shaped_pairs([synth.DATASET_SHAPES["lrssl"]])
enc = G.build_enc_graph(drug, dis, labels, nd, ns, device=dev).int()
torch.manual_seed(0)


def f64_reference(layer, xd, xs, gd, gs):
    """Same layer, dense 0/1 matrices, float64, CPU autograd."""
    att, basis = layer.att.detach().double().cpu().requires_grad_(True), layer.basis.detach().double().cpu().requires_grad_(True)
    W = torch.matmul(att, basis.view(layer.basis_units, -1)).view(-1, layer.user_in_units, layer.msg_units)
    ci_d, ci_s = enc.nodes["drug"].data["ci"].double().cpu(), enc.nodes["disease"].data["ci"].double().cpu()
    out = {"drug": 0, "disease": 0}
    for can in enc.canonical_etypes:
        st, et, dt = can
        r = int(et.replace("rev-", ""))
        rel = enc[can]
        A = torch.zeros(rel.n_dst, rel.n_src, dtype=torch.float64)
        A.index_put_((rel.dst.long().cpu(), rel.src.long().cpu()), torch.ones(rel.number_of_edges(), dtype=torch.float64), accumulate=True)
        x = (xd if st == "drug" else xs).double().cpu()
        cj = (ci_d if st == "drug" else ci_s)  # symm: cj == ci per node type
        out[dt] = out[dt] + A @ (cj * (x @ W[r]))
    yd = torch.nn.functional.leaky_relu(ci_d * out["drug"], 0.1)
    ys = torch.nn.functional.leaky_relu(ci_s * out["disease"], 0.1)
    w, b = layer.ufc.weight.detach().double().cpu(), layer.ufc.bias.detach().double().cpu()
    loss = ((yd @ w.t() + b) * gd.double().cpu()).sum() + ((ys @ w.t() + b) * gs.double().cpu()).sum()
    loss.backward()
    return att.grad, basis.grad


for width, ini in ((1024, True), (128 * 1, False)):
    layer = L.GCMCLayer([0, 1], 768 if ini else 128, 768 if ini else 128, width, 128, dropout_rate=0.0, agg="sum",
                        agg_act=L.get_activation("leaky"), ini=ini, share_user_item_param=True).to(dev)
    fin = 768 if ini else 128
    xd = torch.nn.functional.normalize(torch.randn(nd, fin, device=dev))
    xs = torch.nn.functional.normalize(torch.randn(ns, fin, device=dev))
    gd, gs = torch.randn(nd, 128, device=dev), torch.randn(ns, 128, device=dev)
    ref_att, ref_basis = f64_reference(layer, xd, xs, gd, gs)
    print("== GCMCLayer msg width %d (in %d): f64 basis-grad max %.3e" % (layer.msg_units, fin, float(ref_basis.abs().max())), flush=True)
    for comp in (False, True):
        layer.complement_form = comp
        layer.zero_grad()
        layer.train()
        od, os_ = layer(enc, xd, xs)
        ((od * gd).sum() + (os_ * gs).sum()).backward()
        eb = float((layer.basis.grad.double().cpu() - ref_basis).abs().max() / ref_basis.abs().max())
        ea = float((layer.att.grad.double().cpu() - ref_att).abs().max() / ref_att.abs().max())
        print("   complement=%-5s basis-grad err / max %.2e   att-grad err / max %.2e" % (comp, eb, ea), flush=True)
    child = G.random_edge_dropout(enc, 0.1)
    for graph, gname in ((enc, "un-dropped"), (child, "10% dropout on the fly")):
        for comp in (False, True):
            layer.complement_form = comp

            def fwd():
                with torch.no_grad():
                    layer(graph, xd, xs)

            def fwdbwd():
                layer.zero_grad()
                od, os_ = layer(graph, xd, xs)
                ((od * gd).sum() + (os_ * gs).sum()).backward()

            res = []
            for fn in (fwd, fwdbwd):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    fn()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 100 * 1e6)
            print("   %-24s complement=%-5s layer fwd %.1f us, fwd+bwd %.1f us" % (gname, comp, res[0], res[1]), flush=True)
