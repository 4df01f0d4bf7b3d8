import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from dream_gnn_amd import graph as G
dev = torch.device("cuda:0")
for N, k in ((100_000, 64), (50_000, 64), (100_000, 4)):
    X = torch.randn(N, 768, device=dev)
    for fused in (True, False):
        G.feature_similarity_graph(X, k, fused=fused); torch.cuda.synchronize()
        t0 = time.perf_counter(); A = G.feature_similarity_graph(X, k, fused=fused); torch.cuda.synchronize()
        print("feature_similarity_graph N=%d k=%d fused=%s: %.1f ms, nnz %d" % (N, k, fused, (time.perf_counter() - t0) * 1e3, A._nnz()), flush=True)
