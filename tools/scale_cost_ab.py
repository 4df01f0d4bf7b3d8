"""What the fused diagonal scalings cost the XCD-sliced pair on the config-4 products: src_scale is a random 4-byte
gather per edge (one more cache-line request beside the row's four), dst_scale one multiply in the reduce; and what a
pre-scaled table (one elementwise pass, then the unweighted product) costs instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")


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


F = 128
n_drug, n_dis, E = 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
for name, dst, src, n_dst, n_src in (("drug->disease (51 MB table)", dis, drug, n_dis, n_drug), ("disease->drug (26 MB table)", drug, dis, n_drug, n_dis)):
    X = torch.randn(n_src, F, device=dev)
    ss, ds = synth.degree_norm(src, n_src), synth.degree_norm(dst, n_dst)
    sl = ops.SlicedCSR(dst, src, n_dst, n_src)
    Y = torch.empty(n_dst, F, device=dev)
    Xs = torch.empty_like(X)
    fns = {"src_scale + dst_scale fused (shipped)": lambda: sl.spmm(X, ss, ds, out=Y),
           "dst_scale only": lambda: sl.spmm(X, None, ds, out=Y),
           "no scaling": lambda: sl.spmm(X, None, None, out=Y),
           "pre-scaled table: torch.mul pass + dst_scale only": lambda: (torch.mul(X, ss.view(-1, 1), out=Xs), sl.spmm(Xs, None, ds, out=Y))}
    print("==", name)
    for k, v in timeit(fns).items():
        print("   %-52s %.4f ms" % (k, v), flush=True)
