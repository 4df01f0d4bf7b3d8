"""(f4) the 256 x 256 screen kernels: correctness of sampled rows against a float64 brute force on shapes that cover
K chunk counts 1 / 2 / 3 / 12, both sweeps and the overflow paths, then timing.  DGMI_KNN_SCREEN_V1=1 in the environment
(read once into the library's tuning, also `_lib.set_tuning("knn_screen_first", 1)`) runs the first (register-staged) 256 x 256
kernel instead of the phase-interleaved LDS-DMA one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
which = "v1" if os.environ.get("DGMI_KNN_SCREEN_V1") == "1" else "v8"


def check(N, D, k, clusters=False, seed=0):
    gen = torch.Generator().manual_seed(seed + N + D + k)
    X = torch.randn(N, D, generator=gen)
    if clusters:
        X[3000:3400] = X[3000] + 1e-3 * torch.randn(400, D, generator=gen)
        X[N - 700:N - 100] = X[N - 1] + 1e-3 * torch.randn(600, D, generator=gen)
    xn = (X / X.norm(dim=1, keepdim=True)).to(dev)
    nbr = ops.knn_cosine_topk(xn, k).long()
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(1500), torch.arange(N - 1500, N), torch.randperm(N, generator=gen)[:1500],
                      torch.arange(2900, 3500)]).to(dev)
    sim = xn[rows].double() @ xn.double().t()
    got = torch.gather(sim, 1, nbr[rows])
    want = torch.topk(sim, k, dim=1).values
    err = float((got - want).abs().max())
    distinct = all(len(set(r.tolist())) == k for r in nbr[rows].cpu())
    print("check %s N=%d D=%d k=%d clusters=%s: max err %.2e distinct=%s %s" % (which, N, D, k, clusters, err, distinct,
          "OK" if err <= 2e-6 and distinct else "FAIL"), flush=True)
    return err <= 2e-6 and distinct


def timeit(fn, reps=3, warm=1):
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


ok = True
if "--check" in sys.argv:
    for N, D, k, cl in ((50001, 72, 6, False), (50000, 40, 3, False), (52000, 136, 5, True), (60000, 72, 40, False),
                        (49152, 64, 4, True), (65000, 768, 4, False), (51000, 200, 20, True), (50500, 96, 64, True), (70000, 768, 64, False)):
        ok &= check(N, D, k, cl)
if "--time" in sys.argv:
    for N, D, k in ((100_000, 768, 4), (50_000, 768, 4), (100_000, 768, 64), (50_000, 768, 64)):
        x = torch.randn(N, D, device=dev)
        xn = x / x.norm(dim=1, keepdim=True)
        t = timeit(lambda: ops.knn_cosine_topk(xn, k))
        print("time %s N=%d D=%d k=%d: %.3f ms  (%.0f TFLOP/s of the full rectangle)" % (which, N, D, k, t, 2.0 * N * N * D / t / 1e9), flush=True)
sys.exit(0 if ok else 1)
