import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
spec_path = os.path.join(os.path.dirname(__file__), "model_step_bench.py")
src = open(spec_path).read().split("if os.environ.get(\"ONLY\")")[0]
ns = {"__file__": spec_path}
exec(compile(src, "msb", "exec"), ns)
H, M, G, dev = ns["H"], ns["M"], ns["G"], ns["dev"]
batch, labels, args = ns["problem"](763, 681, 768, 128)
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): H.train_step(net, opt, batch, labels)
print("augment only           %.2f ms" % t(lambda: H.augment(batch)))
print("  enc edge dropout      %.2f ms" % t(lambda: G.random_edge_dropout(batch["enc_graph"], 0.1)))
print("  4x sparse dropout     %.2f ms" % t(lambda: [G.random_edge_dropout_sparse(batch[k], 0.1) for k in ("drug_graph", "disease_graph", "drug_feature_graph", "disease_feature_graph")]))
print("  4x feature noise      %.2f ms" % t(lambda: [batch[k] + torch.randn_like(batch[k]) * 0.05 for k in ("drug_feat", "disease_feat", "drug_sim_feat", "disease_sim_feat")]))
aug = H.augment(batch)
def fb(b):
    loss, _ = H.forward_loss(net, b, labels, 0.1); opt.zero_grad(); loss.backward()
print("fwd+bwd on fixed batch  %.2f ms" % t(lambda: fb(batch)))
print("fwd+bwd on an augmented batch (same object reused) %.2f ms" % t(lambda: fb(aug)))
print("fwd+bwd on a fresh augmented batch each time %.2f ms" % t(lambda: fb(H.augment(batch))))
net.train()
print("full step aug           %.2f ms" % t(lambda: H.train_step(net, opt, batch, labels)))
print("full step no aug        %.2f ms" % t(lambda: H.train_step(net, opt, batch, labels, do_augment=False)))
from dream_gnn_amd import layers as L
def views():
    a = H.augment(batch)
    for nt in ("drug", "disease"):
        csr, _ = a["enc_graph"].fused_relations(nt)
        csr.vals; csr.transposed()
    for k in ("drug_graph", "disease_graph", "drug_feature_graph", "disease_feature_graph"):
        g = L.adjacency_csr(a[k]); g.vals; g.transposed()
print("augment + all value views %.2f ms" % t(views))
def views_enc():
    a = G.random_edge_dropout(batch["enc_graph"], 0.1)
    for nt in ("drug", "disease"):
        csr, _ = a.fused_relations(nt)
        csr.vals; csr.transposed()
print("  enc dropout + fused views %.2f ms" % t(views_enc))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): fb(H.augment(batch))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
