"""(r4) The drop-in MODULES at config-4 scale: 3 GCMCLayers (128 -> 128 -> 128, two ratings split 97 % / 3 % as SURVEY 8(d)
suggests) on the 100k x 50k / 10 M-edge graph + a 2-layer GCN on the kNN-64 graph, forward + backward, with the
reference's per-step edge dropout (train.py:267) — against the sum of the products they run.  What the layouts'
compaction, the pre-scale passes and the autograd plumbing cost around the SpMM kernels at scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import graph as G, layers as L, ops, synth

dev = torch.device("cuda:0")
ND, NS, E, F = 100_000, 50_000, 10_000_000, 128
drug, dis = synth.bipartite_edges(ND, NS, E, seed=0, device=dev)
labels = (torch.rand(E, device=dev) < 0.03).float()
enc = G.build_enc_graph(drug, dis, labels, ND, NS, device=dev).int()
r, c, v = synth.knn_sim_graph(ND, 64, 21, dev)
adj = torch.sparse_coo_tensor(torch.stack([r.long(), c.long()]), v, (ND, ND))
adj._dgmi_trusted = True
torch.manual_seed(0)
gcmc = torch.nn.ModuleList([L.GCMCLayer([0, 1], F, F, F, F, dropout_rate=0.1, agg="sum", agg_act="leaky", ini=False,
                                        share_user_item_param=True, device=None) for _ in range(3)]).to(dev)
gcn = L.GCN(F, F, F, 0.1).to(dev)
x_drug = torch.randn(ND, F, device=dev, requires_grad=True)
x_dis = torch.randn(NS, F, device=dev, requires_grad=True)


def step(dropout):
    g = G.random_edge_dropout(enc, 0.1) if dropout else enc
    a = G.random_edge_dropout_sparse_views([adj], 0.1)[0] if dropout else adj
    d, s = x_drug, x_dis
    for layer in gcmc:
        d, s = layer(g, d, s)
    y = gcn(x_drug, a)
    (d.sum() + s.sum() + y.sum()).backward()


def wall(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for drop in (False, True):
    gcmc.train()
    gcn.train()
    print("3 GCMC layers + 2-layer GCN, fwd + bwd, edge dropout %s: %.3f ms per step" % ("on " if drop else "off", wall(lambda: step(drop))), flush=True)

# hidden host synchronisation in the per-step path at scale (a readback per step would serialise host and device)
import warnings
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(2):
        step(True)
torch.cuda.set_sync_debug_mode("default")
msgs = ["%s @ %s:%d" % (str(x.message)[:90], x.filename.split("/")[-1], x.lineno) for x in w if "synchron" in str(x.message).lower() and "prototype" not in str(x.message)]
print("synchronising calls in 2 dropped steps: %d" % len(msgs))
for m in sorted(set(msgs)):
    print("  ", msgs.count(m), "x", m)
