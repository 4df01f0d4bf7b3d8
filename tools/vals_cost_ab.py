"""What the per-edge value stream costs the XCD-sliced pair on the config-4 kNN-64 products (u_mul_e -> sum): the same
graph with and without values, and with the degree spread of the symmetrised kNN graph priced against a regular graph of
the same size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth


def timeit(fns, rounds=15, inner=5):
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(b) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


dev = torch.device("cuda:0")
F = 128
for n in (100_000, 50_000):
    r, c, v = synth.knn_sim_graph(n, 64, 2, dev)
    X = torch.randn(n, F, device=dev)
    sl = ops.SlicedCSR(r, c, n, n, vals=v)
    sl0 = ops.SlicedCSR(r, c, n, n)
    E = r.numel()
    g = torch.Generator(device=dev).manual_seed(5)
    rr = torch.arange(n, device=dev, dtype=torch.int32).repeat_interleave(E // n)
    cc = torch.randint(0, n, (rr.numel(),), generator=g, device=dev, dtype=torch.int32)
    reg = ops.SlicedCSR(rr, cc, n, n)
    Y = torch.empty(n, F, device=dev)
    deg = torch.bincount(r.long(), minlength=n)
    print("== kNN-64, n = %d, nnz = %d, degree min/mean/max = %d / %.1f / %d" % (n, E, deg.min(), deg.float().mean(), deg.max()))
    fns = {"weighted (shipped)": lambda: sl.spmm(X, None, None, out=Y),
           "same graph, no values": lambda: sl0.spmm(X, None, None, out=Y),
           "regular graph, %d edges per row, no values" % (E // n): lambda: reg.spmm(X, None, None, out=Y)}
    for k, t in timeit(fns).items():
        print("   %-52s %.4f ms" % (k, t), flush=True)
