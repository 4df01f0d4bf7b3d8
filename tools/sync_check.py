import os, sys, warnings
sys.path.insert(0, '/root/repo')
import torch
from dream_gnn_amd import graph as G, harness as H, model as M, synth, layers as L
dev = torch.device("cuda:0")
batch, labels = synth.dataset_shaped_batch([synth.DATASET_SHAPES["lrssl"]], emb=768, k=4, seed=0, device=dev)
args = synth.net_args(out_units=128, n_drug=batch["n_drug"], n_dis=batch["n_dis"])
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
for _ in range(3):
    H.train_step(net, opt, batch, labels)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(2):
        H.train_step(net, opt, batch, labels)
torch.cuda.set_sync_debug_mode("default")
msgs = [str(x.message)[:100] + " @ " + "%s:%d" % (x.filename.split('/')[-1], x.lineno) for x in w if "synchron" in str(x.message).lower()]
print("lrssl-shaped eager train_step: %d synchronising calls in 2 steps" % len(msgs))
for m in sorted(set(msgs)): print("  ", m, msgs.count(m))
