"""In-process A/B of the lrssl-shaped training step (same box, interleaved repeats): fused epilogue
on/off, dense path on/off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import harness as H, layers as L, model as M, ops, synth

dev = torch.device("cuda:0")
batch, labels = synth.dataset_shaped_batch([synth.DATASET_SHAPES["lrssl"]], emb=768, k=4, seed=0, device=dev)
args = synth.net_args(n_drug=batch["n_drug"], n_dis=batch["n_dis"])
torch.manual_seed(0)
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)


MODE = os.environ.get("MODE", "train")


def one():
    if MODE == "train":
        H.train_step(net, opt, batch, labels)
    elif MODE == "train_noaug":
        H.train_step(net, opt, batch, labels, do_augment=False)
    else:
        with torch.no_grad():
            net.eval()
            H.forward_loss(net, batch, labels, 0.1)


def run(steps=40):
    for _ in range(5):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


variants = {"epilogue fused, dense on": (True, 0.25), "epilogue separate, dense on": (False, 0.25),
            "epilogue fused, dense off": (True, 2.0), "epilogue separate, dense off": (False, 2.0)}
res = {k: [] for k in variants}
for rep in range(4):
    for name, (fe, dens) in variants.items():
        L.GCMCLayer.fuse_epilogue = fe
        ops.DENSE_MIN_DENSITY = dens
        res[name].append(run())
for name, v in res.items():
    print(MODE, "%-32s %s  median %.3f ms" % (name, " ".join("%.3f" % x for x in v), sorted(v)[len(v) // 2]), flush=True)
