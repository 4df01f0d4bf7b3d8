"""A/B in one process of the sliced pair's chunk pipeline (DGMI_SLICED_OVERLAP = chunks, DGMI_SLICED_REDUCE_BLOCKS =
thin reduce grid): the reduce of chunk c under the gather of chunk c + 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth
dev = torch.device("cuda:0")

def timeit(fns, rounds=12, inner=5):
    for f in fns.values():
        for _ in range(3): f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner): f()
            b.record(); torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(b) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}

def with_env(fn, **env):
    def run():
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try: return fn()
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    return run

F = 128
n_drug, n_dis, E = 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
r, c, v = synth.knn_sim_graph(n_drug, 64, 2, dev)
for name, dst, src, n_dst, n_src, vals in (("drug->disease (51 MB table, 50k rows)", dis, drug, n_dis, n_drug, None),
                                           ("disease->drug (26 MB table, 100k rows)", drug, dis, n_drug, n_dis, None),
                                           ("drug kNN-64 weighted (100k rows)", r, c, n_drug, n_drug, v)):
    X = torch.randn(n_src, F, device=dev)
    ss = None if vals is not None else synth.degree_norm(src, n_src)
    ds = None if vals is not None else synth.degree_norm(dst, n_dst)
    sl = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals)
    Y = torch.empty(n_dst, F, device=dev)
    ref = with_env(lambda: sl.spmm(X, ss, ds), DGMI_SLICED_OVERLAP="1")()
    fns = {"serial (1 chunk)": with_env(lambda: sl.spmm(X, ss, ds, out=Y), DGMI_SLICED_OVERLAP="1")}
    for C in ("2", "4", "8"):
        for rb in ("128", "256", "512", "1024"):
            key = "%s chunks, thin reduce %s blocks" % (C, rb)
            fns[key] = with_env(lambda: sl.spmm(X, ss, ds, out=Y), DGMI_SLICED_OVERLAP=C, DGMI_SLICED_REDUCE_BLOCKS=rb)
            y = fns[key]()
            assert torch.equal(y, ref), key  # same kernels on row ranges: bit-identical
    print("==", name, flush=True)
    for k, t in timeit(fns).items():
        print("   %-36s %.4f ms" % (k, t), flush=True)
