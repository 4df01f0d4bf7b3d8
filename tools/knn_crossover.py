"""(f4) where the bf16-screened path overtakes the fp32 kernel: run with DGMI_KNN_SCREEN_MIN_ROWS=1 and =1000000000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
for N in (1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288):
    for k in (4, 16):
        x = torch.randn(N, 768, device=dev)
        xn = x / x.norm(dim=1, keepdim=True)
        for _ in range(3):
            ops.knn_cosine_topk(xn, k)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.knn_cosine_topk(xn, k)
        b.record()
        torch.cuda.synchronize()
        print("min_rows=%s N=%6d k=%2d: %.3f ms" % (os.environ.get("DGMI_KNN_SCREEN_MIN_ROWS", "default"), N, k, a.elapsed_time(b) / 20), flush=True)
