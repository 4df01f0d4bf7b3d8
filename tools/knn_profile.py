"""(f4) one screened kNN search per (N, k) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x = torch.randn(N, 768, device=dev)
xn = x / x.norm(dim=1, keepdim=True)
for _ in range(3):
    ops.knn_cosine_topk(xn, k)
torch.cuda.synchronize()
