import os, sys
sys.path.insert(0, "/root/repo")
import torch
from dream_gnn_amd import ops, synth
dev = torch.device("cuda:0")
F = 128
n_drug, n_dis, E = 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
X = torch.randn(n_drug, F, device=dev)
ss = synth.degree_norm(drug, n_drug); ds = synth.degree_norm(dis, n_dis)
sl = ops.SlicedCSR(dis, drug, n_dis, n_drug)
Y = torch.empty(n_dis, F, device=dev)
def timed(stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(5): sl.spmm(X, ss, ds, out=Y)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): sl.spmm(X, ss, ds, out=Y)
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
side = torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)
for _ in range(3):
    print("default %.4f  side %.4f  high-priority side %.4f" % (timed(torch.cuda.default_stream()), timed(side), timed(hi)), flush=True)
