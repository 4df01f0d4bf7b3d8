"""A/B of the row-owned slice-swept kernel against the XCD-sliced pair and the planned kernel on
config 4's products (10 M edges, F = 128), sweeping blocks per CU and slice count.
    python tools/owned_bench.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")
F = 128
ND, NS, E = 100_000, 50_000, 10_000_000


def timeit(fn, reps=30, warm=5):
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


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    drug, dis = synth.bipartite_edges(ND, NS, E, seed=0, device=dev)
    cj, ci = synth.degree_norm(drug, ND), synth.degree_norm(dis, NS)
    g = torch.Generator(device=dev).manual_seed(3)
    xd, xs = torch.randn(ND, F, generator=g, device=dev), torch.randn(NS, F, generator=g, device=dev)
    cases = [("drug->disease (51 MB table, 50k rows)", dis, drug, NS, ND, xd, cj, ci),
             ("disease->drug (26 MB table, 100k rows)", drug, dis, ND, NS, xs, ci, cj)]
    r, c, v = synth.knn_sim_graph(ND, 64, 21, dev)
    cases.append(("drug kNN-64 weighted (51 MB table, 100k rows)", r, c, ND, ND, xd, None, None, v))
    for case in cases:
        name, dst, src, n_dst, n_src, X, ss, ds = case[:8]
        vals = case[8] if len(case) > 8 else None
        print("==", name, flush=True)
        base = ops.CSRGraph(dst, src, n_dst, n_src, vals=vals)
        y_ref = ops.spmm_csr_raw(base.indptr, base.indices, base.vals, X, ss, ds, plan=base.plan)
        t_plan = timeit(lambda: ops.spmm_csr_raw(base.indptr, base.indices, base.vals, X, ss, ds, plan=base.plan, out=y_ref))
        sl = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals)
        out = torch.empty_like(y_ref)
        t_sl = timeit(lambda: sl.spmm(X, ss, ds, out=out))
        err = float((out - y_ref).abs().max() / y_ref.abs().max())
        print("  planned %.4f ms   xcd-sliced pair %.4f ms (rel diff %.1e)" % (t_plan, t_sl, err), flush=True)
        sweep = [(0, 0)] if quick else [(m, s) for m in (5, 4) for s in (16, 32)]
        paces = [None, (1, 75), (1, 50), (1, 90), (2, 75), (2, 50)]
        for m, s in sweep:
            ow = ops.OwnedCSR(dst, src, n_dst, n_src, F=F, vals=vals, blocks_per_cu=m, n_slices=s)
            gm = ow.geom
            line = "  owned m=%d S=%2d (rmax %d, rounds %d, slice %.1f MB):" % (gm.blocks // 256, gm.n_slices, gm.rmax, gm.rounds, gm.slice_width * F * 4 / 1e6)
            for pace in paces:
                ow.paced = pace is not None
                if pace is not None:
                    os.environ["DGMI_OWNED_PACE_LAG"], os.environ["DGMI_OWNED_PACE_PCT"] = str(pace[0]), str(pace[1])
                t = timeit(lambda: ow.spmm(X, ss, ds, out=out))
                err = float((out - y_ref).abs().max() / y_ref.abs().max())
                assert err < 1e-5, err
                line += "  %s %.4f" % ("free" if pace is None else "lag%d/%d%%" % pace, t)
            print(line + "   [sliced %.4f]" % t_sl, flush=True)
            del ow


if __name__ == "__main__":
    main()
