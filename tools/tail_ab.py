"""A/B of the XCD-sliced product's two endings on the config-4 products: 8 slices + reduce_planes_kernel (8 planes
written and read back, then nothing else to do) against 9 slices where the launch that adds the 8 planes also gathers
the 9th slice (DGMI_SLICED_TAIL=1).  Timed in a loop of its own and behind a cache flush (as inside a step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")
flush_buf = torch.empty(160 << 20, dtype=torch.float32, device=dev)  # 640 MB: more than the Infinity Cache


def timeit(fns, rounds=12, inner=5, flush=False):
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            if flush:
                for _ in range(inner):
                    flush_buf.add_(1.0)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    f()
                    b.record()
                    torch.cuda.synchronize()
                    ts[k].append(a.elapsed_time(b))
            else:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(inner):
                    f()
                b.record()
                torch.cuda.synchronize()
                ts[k].append(a.elapsed_time(b) / inner)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in ts.items()}


def with_env(fn, **env):
    def run():
        old = {k: os.environ.get(k) for k in env}
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        try:
            return fn()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return run


F = 128
n_drug, n_dis, E = 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
r, c, v = synth.knn_sim_graph(n_drug, 64, 2, dev)
r2, c2, v2 = synth.knn_sim_graph(n_dis, 64, 2, dev)
cases = [("drug->disease (51 MB table, 50k rows)", dis, drug, n_dis, n_drug, None),
         ("disease->drug (26 MB table, 100k rows)", drug, dis, n_drug, n_dis, None),
         ("drug kNN-64 weighted (51 MB, 100k rows)", r, c, n_drug, n_drug, v),
         ("disease kNN-64 weighted (26 MB, 50k rows)", r2, c2, n_dis, n_dis, v2)]
only = os.environ.get("TAIL_AB_CASES")
for ci, (name, dst, src, n_dst, n_src, vals) in enumerate(cases):
    if only and str(ci) not in only.split(","):
        continue
    X = torch.randn(n_src, F, device=dev)
    ss = None if vals is not None else synth.degree_norm(src, n_src)
    ds = None if vals is not None else synth.degree_norm(dst, n_dst)
    s8 = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals)
    Y8, Y9 = torch.empty(n_dst, F, device=dev), torch.empty(n_dst, F, device=dev)
    ref = s8.spmm(X, ss, ds)
    fns = {"8 slices + reduce (shipped)": lambda: s8.spmm(X, ss, ds, out=Y8)}
    keepalive = []
    for frac in (1.0 / 9, 0.15, 0.2, 0.25, 0.3, 0.4, 0.5):
        w = str(int(-(-n_src * (1.0 - frac) // 8)))
        s9 = with_env(lambda: ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals, n_slices=9), DGMI_SLICE_WIDTH=w)()
        keepalive.append(s9)
        y9 = with_env(lambda: s9.spmm(X, ss, ds), DGMI_SLICED_TAIL="1", DGMI_SLICE_WIDTH=w)()
        err = float((y9 - ref).abs().max())
        for rows in ("1", "2", "4"):
            fns["tail %.2f of the sources (width %s, max|d| %.1e): rows %s" % (frac, w, err, rows)] = with_env(
                lambda s9=s9: s9.spmm(X, ss, ds, out=Y9), DGMI_SLICED_TAIL="1", DGMI_TAIL_ROWS=rows, DGMI_SLICE_WIDTH=w)
    warm = timeit(fns)
    cold = timeit(fns, rounds=3, inner=5, flush=True)
    print("== %s" % name)
    for k in fns:
        print("   %-66s loop %.4f ms (min %.4f)   behind a flush %.4f ms" % (k, warm[k][0], warm[k][1], cold[k][0]), flush=True)
