"""(f4) triangular sweep of the screen (DGMI_KNN_SYM=1) against the full rectangle (=0), by N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
for N in (2048, 4096, 8192, 16384, 20000, 32768, 50000, 100000):
    for k in (4, 16):
        x = torch.randn(N, 768, device=dev)
        xn = x / x.norm(dim=1, keepdim=True)
        reps = 10 if N <= 32768 else 3
        for _ in range(2):
            ops.knn_cosine_topk(xn, k)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ops.knn_cosine_topk(xn, k)
        b.record()
        torch.cuda.synchronize()
        print("SYM=%s N=%6d k=%2d: %.3f ms" % (os.environ.get("DGMI_KNN_SYM", "default"), N, k, a.elapsed_time(b) / reps), flush=True)
