"""In-process A/B of the XCD-sliced pair on the config-4 products: 32-bit row offsets on / off, and VERDICT r2
item 1(a) — 4 planes (two XCDs per source slice) with narrower column groups — against the shipped 8 planes."""
import os
import sys

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
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in ts.items()}


_KNOBS = {"DGMI_SLICED_LPR": "sliced_lpr", "DGMI_NO_OFF32": "sliced_no_off32", "DGMI_SLICED_ROWS": "sliced_rows"}


def with_env(fn, **env):
    """Launch-parameter overrides go through dgmi_set_tuning (the library reads its environment once, not per launch);
    anything else is set in os.environ as before."""
    from dream_gnn_amd import _lib

    knobs = {_KNOBS[k]: int(v or 0) for k, v in env.items() if k in _KNOBS}
    env = {k: v for k, v in env.items() if k not in _KNOBS}

    def run():
        for k, v in knobs.items():
            _lib.set_tuning(k, v)
        try:
            return run_env()
        finally:
            for k in knobs:
                _lib.set_tuning(k, 0)

    def run_env():
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
cases = [("drug->disease (51 MB table, 50k rows)", dis, drug, n_dis, n_drug, None, ("16", "8")),
         ("disease->drug (26 MB table, 100k rows)", drug, dis, n_drug, n_dis, None, ("32", "16")),
         ("drug kNN-64 weighted (51 MB, 100k rows)", r, c, n_drug, n_drug, v, ("16", "8"))]
for name, dst, src, n_dst, n_src, vals, (lpr8, lpr4) in cases:
    X = torch.randn(n_src, F, device=dev)
    ss = None if vals is not None else synth.degree_norm(src, n_src)
    ds = None if vals is not None else synth.degree_norm(dst, n_dst)
    s8 = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals)
    s4 = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals, n_slices=4)
    Y8, Y4 = torch.empty(n_dst, F, device=dev), torch.empty(n_dst, F, device=dev)
    ref = s8.spmm(X, ss, ds)
    y4 = with_env(lambda: s4.spmm(X, ss, ds), DGMI_SLICED_LPR=lpr4)()
    print("== %s: 4-plane vs 8-plane max|d| %.2e (max|y| %.2e)" % (name, float((y4 - ref).abs().max()), float(ref.abs().max())), flush=True)
    fns = {"8 planes, lpr %s (shipped rule), off32" % lpr8: lambda: s8.spmm(X, ss, ds, out=Y8),
           "8 planes, lpr %s, 64-bit addresses" % lpr8: with_env(lambda: s8.spmm(X, ss, ds, out=Y8), DGMI_NO_OFF32="1"),
           "4 planes, lpr %s (2 XCDs per slice)" % lpr4: with_env(lambda: s4.spmm(X, ss, ds, out=Y4), DGMI_SLICED_LPR=lpr4),
           "4 planes, lpr %s (slices 2x an L2)" % lpr8: with_env(lambda: s4.spmm(X, ss, ds, out=Y4), DGMI_SLICED_LPR=lpr8)}
    for k, (med, mn) in timeit(fns).items():
        print("   %-44s median %.4f ms  min %.4f" % (k, med, mn), flush=True)
