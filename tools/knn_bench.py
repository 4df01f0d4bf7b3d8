"""(f4) fused MFMA kNN kernel vs the row-blocked torch GEMM + top-k path (neighbour search only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def torch_path(xn, k, block_rows=8192):
    n = xn.shape[0]
    out = torch.empty((n, k), dtype=torch.int64, device=xn.device)
    for lo in range(0, n, block_rows):
        hi = min(lo + block_rows, n)
        out[lo:hi] = torch.topk(xn[lo:hi] @ xn.t(), k, dim=1).indices
    return out


for N, D, k in ((763, 768, 4), (1256, 768, 4), (8192, 768, 4), (20_000, 768, 4), (100_000, 768, 4), (100_000, 768, 16), (20_000, 768, 64), (100_000, 768, 64)):
    x = torch.randn(N, D, device=dev)
    xn = x / x.norm(dim=1, keepdim=True)
    reps = 20 if N < 5000 else 2
    t_f = timeit(lambda: ops.knn_cosine_topk(xn, k), reps=reps)
    t_t = timeit(lambda: torch_path(xn, k), reps=reps)
    flops = 2.0 * N * N * D
    print("N=%6d D=%d k=%2d: fused %.3f ms (%.1f TFLOP/s)   torch GEMM+topk %.3f ms   (%.2fx)"
          % (N, D, k, t_f, flops / t_f / 1e9, t_t, t_t / t_f), flush=True)
