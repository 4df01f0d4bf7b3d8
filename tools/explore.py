"""Scratch experiments on the GPU box (not judged): HBM-bound gather (cfg-5 shard shape),
degree skew, lrssl-shaped small graphs, other feature widths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts)//2]

def report(name, g, X, F, weighted=False, ss=None, ds=None):
    Y = torch.empty(g.n_dst, F, device=dev)
    ms0 = timeit(lambda: ops.spmm_csr_raw(g.indptr, g.indices, g.vals, X, ss, ds, out=Y))
    ms = timeit(lambda: g.spmm(X, ss, ds, out=Y))
    name = name + " [unplanned %.1f us, chunk %d]" % (ms0 * 1e3, g.plan.chunk)
    nnz = g.nnz
    b = nnz*(4*F+4+(4 if weighted else 0)) + g.n_dst*4*F
    print(f"{name}: nnz {nnz/1e6:.2f}M rows {g.n_dst} F {F}: {ms*1e3:.1f} us  {nnz/ms/1e6:.2f} Gedge/s  {b/ms/1e6:.0f} GB/s alg", flush=True)

which = sys.argv[1:] or ["hbm", "skew", "small", "widths"]
gen = torch.Generator(device=dev).manual_seed(0)
if "hbm" in which:
    for n_src in (100_000, 400_000, 800_000, 1_600_000):
        n_dst, E = 50_000, 10_000_000
        dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        g = ops.CSRGraph(dst, src, n_dst, n_src)
        X = torch.randn(n_src, 128, device=dev)
        report(f"gather table {n_src*512/1e6:.0f} MB", g, X, 128)
        del g, X, dst, src
if "skew" in which:
    n_dst, n_src, E = 50_000, 100_000, 10_000_000
    for alpha in (0.0, 0.8, 1.2):
        p = 1.0 / torch.arange(1, n_dst + 1, device=dev, dtype=torch.float64) ** alpha
        dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=gen).to(torch.int32)
        dst = dst[torch.randperm(E, device=dev, generator=gen)]
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        g = ops.CSRGraph(dst, src, n_dst, n_src)
        deg = (g.indptr[1:] - g.indptr[:-1])
        X = torch.randn(n_src, 128, device=dev)
        report(f"zipf alpha {alpha} (max deg {int(deg.max())}, median {int(deg.median())})", g, X, 128)
        del g, X
if "small" in which:
    nd, ns = 763, 681
    pairs = torch.cartesian_prod(torch.arange(nd), torch.arange(ns)).to(dev)
    keep = torch.rand(pairs.shape[0], device=dev, generator=gen) < 0.9
    pairs = pairs[keep]
    lab = torch.rand(pairs.shape[0], device=dev, generator=gen) < 0.006
    for nm, sel in (("rel0", ~lab), ("rel1", lab)):
        d, s = pairs[sel, 0].int(), pairs[sel, 1].int()
        g = ops.CSRGraph(s, d, ns, nd)
        for F in (341, 344, 128, 256):
            X = torch.randn(nd, F, device=dev)
            report(f"lrssl-shape {nm} drug->disease", g, X, F)
    r, c, v = synth.knn_sim_graph(nd, 4, 1, dev)
    g = ops.CSRGraph(r, c, nd, nd, vals=v)
    for F in (768, 128):
        report("lrssl-shape knn-4", g, torch.randn(nd, F, device=dev), F, weighted=True)
    # launch overhead floor: empty graph
    g0 = ops.CSRGraph(torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), 4, 4)
    report("empty (launch floor)", g0, torch.randn(4, 128, device=dev), 128)
if "widths" in which:
    n_dst, n_src, E = 50_000, 100_000, 10_000_000
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
    g = ops.CSRGraph(dst, src, n_dst, n_src)
    for F in (32, 64, 128, 256, 341, 344, 512, 768):
        X = torch.randn(n_src, F, device=dev)
        report("uniform 10M", g, X, F)
        del X
if "decoder" in which:
    for nd, ns, E, tag in ((763, 681, 467_000, "lrssl-shape"), (100_000, 50_000, 10_000_000, "cfg4")):
        s = torch.randint(0, nd, (E,), generator=gen, device=dev, dtype=torch.int32)
        d = torch.randint(0, ns, (E,), generator=gen, device=dev, dtype=torch.int32)
        A, B = torch.randn(nd, 128, device=dev), torch.randn(ns, 128, device=dev)
        out = torch.empty(E, 256, device=dev)
        ms = timeit(lambda: ops.gather_concat_raw(s, d, A, B, out=out))
        ms_t = timeit(lambda: torch.cat([A.index_select(0, s.long()), B.index_select(0, d.long())], 1))
        pairs = ops.EdgePairs(s, d, nd, ns)
        dO = torch.randn(E, 256, device=dev)
        gs, gd = pairs.by_src(), pairs.by_dst()
        ms_b = timeit(lambda: (gs.spmm(dO[:, :128]), gd.spmm(dO[:, 128:])))
        da = torch.zeros(nd, 128, device=dev); db = torch.zeros(ns, 128, device=dev)
        ms_bt = timeit(lambda: (da.index_add_(0, s.long(), dO[:, :128]), db.index_add_(0, d.long(), dO[:, 128:])))
        print(f"decoder gather-concat {tag} E={E}: hip {ms*1e3:.1f} us ({E*256*4*2/ms/1e6:.0f} GB/s r+w) vs torch index_select+cat {ms_t*1e3:.1f} us | bwd hip {ms_b*1e3:.1f} us vs torch index_add {ms_bt*1e3:.1f} us", flush=True)
if "xcdlocal" in which:
    # Upper bound for an XCD-sliced gather: rows whose block lands on XCD s (block = row/4, XCD = block % 8)
    # only reference sources inside slice s of X. Same kernel, same bytes; only L2 locality changes.
    for n_src in (50_000, 100_000, 400_000):
        n_dst, E = 50_000, 10_000_000
        dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        w = n_src // 8
        off = torch.randint(0, w, (E,), generator=gen, device=dev, dtype=torch.int32)
        src_local = ((dst // 4) % 8) * w + off
        src_rand = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        X = torch.randn(n_src, 128, device=dev)
        for nm, src in (("xcd-local", src_local), ("uniform", src_rand)):
            g = ops.CSRGraph(dst, src, n_dst, n_src)
            report(f"{nm} sources, table {n_src*512/1e6:.0f} MB (slice {n_src*64/1e6:.1f} MB)", g, X, 128)
if "sliced" in which:
    for n_dst, n_src, E, weighted in ((50_000, 100_000, 10_000_000, False), (100_000, 50_000, 10_000_000, False), (100_000, 100_000, 12_900_000, True)):
        dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        vals = torch.rand(E, generator=gen, device=dev) if weighted else None
        g = ops.CSRGraph(dst, src, n_dst, n_src, vals=vals)
        X = torch.randn(n_src, 128, device=dev)
        ss = torch.rand(n_src, device=dev); ds = torch.rand(n_dst, device=dev)
        y0 = g.spmm(X, ss, ds)
        for S in (4, 8, 16):
            sl = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals, n_slices=S)
            y1 = sl.spmm(X, ss, ds)
            err = float((y1 - y0).abs().max() / y0.abs().max())
            Y = torch.empty_like(y0)
            ms = timeit(lambda: sl.spmm(X, ss, ds, out=Y))
            ms0 = timeit(lambda: g.spmm(X, ss, ds, out=Y))
            print(f"sliced S={S}: {n_src}->{n_dst} E={E} w={weighted}: sliced {ms*1e3:.1f} us vs planned {ms0*1e3:.1f} us  ({E/ms/1e6:.1f} Gedge/s) rel diff {err:.2e}", flush=True)
if "slicedcap" in which:
    n_dst, E = 50_000, 10_000_000
    for n_src in (12_500, 25_000, 50_000, 100_000, 150_000, 200_000, 300_000, 400_000, 800_000):
        dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        g = ops.CSRGraph(dst, src, n_dst, n_src)
        X = torch.randn(n_src, 128, device=dev)
        Y = torch.empty(n_dst, 128, device=dev)
        sl = ops.SlicedCSR(dst, src, n_dst, n_src)
        ms_s = timeit(lambda: sl.spmm(X, out=Y))
        ms_p = timeit(lambda: ops.spmm_csr_raw(g.indptr, g.indices, None, X, out=Y, plan=g.plan))
        print(f"table {n_src*512/1e6:6.1f} MB: sliced {ms_s*1e3:6.1f} us  planned {ms_p*1e3:6.1f} us  ratio {ms_p/ms_s:.2f}", flush=True)
        del g, sl, X
    # average degree sweep at the 51 MB table
    n_src = 100_000
    X = torch.randn(n_src, 128, device=dev)
    for n_dst in (25_000, 100_000, 400_000, 1_000_000):
        dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        g = ops.CSRGraph(dst, src, n_dst, n_src)
        Y = torch.empty(n_dst, 128, device=dev)
        sl = ops.SlicedCSR(dst, src, n_dst, n_src)
        ms_s = timeit(lambda: sl.spmm(X, out=Y))
        ms_p = timeit(lambda: ops.spmm_csr_raw(g.indptr, g.indices, None, X, out=Y, plan=g.plan))
        print(f"avg degree {E/n_dst:6.1f}: sliced {ms_s*1e3:6.1f} us  planned {ms_p*1e3:6.1f} us  ratio {ms_p/ms_s:.2f}", flush=True)
        del g, sl
if "torchref" in which:
    # What the reference's own call (th.spmm on a sparse COO, layers.py:312) costs on this GPU through
    # ATen/rocSPARSE, next to the HIP path, on the config-4 kNN-64 graph and the bipartite slice.
    n = 100_000
    r, c, v = synth.knn_sim_graph(n, 64, 2, dev)
    X = torch.randn(n, 128, device=dev)
    adj = torch.sparse_coo_tensor(torch.stack([r.long(), c.long()]), v, (n, n)).coalesce()
    csr = adj.to_sparse_csr()
    g = ops.CSRGraph(r, c, n, n, vals=v)
    Y = torch.empty(n, 128, device=dev)
    t_coo = timeit(lambda: torch.spmm(adj, X), n=10)
    t_csr = timeit(lambda: csr @ X, n=10)
    t_hip = timeit(lambda: g.spmm(X, out=Y))
    print(f"kNN-64 N=100k nnz={g.nnz}: torch.spmm(COO) {t_coo*1e3:.0f} us | torch CSR @ {t_csr*1e3:.0f} us | HIP path {t_hip*1e3:.0f} us", flush=True)
    xr = X.clone().requires_grad_(True)
    def fb():
        y = torch.spmm(adj, xr); y.backward(Y)
    def fb_hip():
        y = ops.spmm_csr(g, xr); y.backward(Y)
    g.transposed()
    print(f"   fwd+bwd: torch.spmm(COO) {timeit(fb, n=10)*1e3:.0f} us | HIP path {timeit(fb_hip)*1e3:.0f} us", flush=True)
    nd, ns, E = 100_000, 50_000, 10_000_000
    drug, dis = synth.bipartite_edges(nd, ns, E, 0, dev)
    cj, ci = synth.degree_norm(drug, nd), synth.degree_norm(dis, ns)
    Xd = torch.randn(nd, 128, device=dev)
    gb = ops.CSRGraph(dis, drug, ns, nd)
    Yb = torch.empty(ns, 128, device=dev)
    def gcmc_torch():  # the update_all(copy_u, sum) stand-in a torch-only port would use
        h = Xd * cj[:, None]
        out = torch.zeros(ns, 128, device=dev).index_add_(0, dis.long(), h.index_select(0, drug.long()))
        return out * ci[:, None]
    t_t = timeit(gcmc_torch, n=5)
    t_h = timeit(lambda: gb.spmm(Xd, cj, ci, out=Yb))
    print(f"GCMC slice 10M edges: torch index_select+index_add_ {t_t*1e3:.0f} us | HIP path {t_h*1e3:.0f} us", flush=True)
if "widths2" in which:
    n_dst, n_src, E = 50_000, 100_000, 10_000_000
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
    g = ops.CSRGraph(dst, src, n_dst, n_src)
    for F in (32, 64, 128, 256, 344, 512):
        X = torch.randn(n_src, F, device=dev)
        Y = torch.empty(n_dst, F, device=dev)
        ms = timeit(lambda: g.spmm(X, out=Y))
        ms_p = timeit(lambda: ops.spmm_csr_raw(g.indptr, g.indices, None, X, out=Y, plan=g.plan))
        print(f"F={F}: auto {ms*1e3:.0f} us ({'sliced' if g._use_sliced(F, n_dst, n_src, g.regular) else 'planned'}) planned {ms_p*1e3:.0f} us  {E*(4*F+4)/ms/1e6:.0f} GB/s alg", flush=True)
if "buildtimes" in which:
    n_dst, n_src, E = 50_000, 100_000, 10_000_000
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
    print("csr_from_coo (plain) %.3f ms" % timeit(lambda: ops.csr_from_coo(dst, src, n_dst), n=10))
    print("SlicedCSR build      %.3f ms" % timeit(lambda: ops.SlicedCSR(dst, src, n_dst, n_src), n=10))
    ip, _, _ = ops.csr_from_coo(dst, src, n_dst)
    print("plan build           %.3f ms" % timeit(lambda: ops.build_plan(ip, E), n=10))
    print("CSRGraph (validated: plain + plan + readback) %.3f ms" % timeit(lambda: ops.CSRGraph(dst, src, n_dst, n_src), n=10))
    print("random_subset_mask   %.3f ms" % timeit(lambda: ops.random_subset_mask(E, 9_000_000, 7, dev), n=10))
    print("torch.randperm(E)    %.3f ms" % timeit(lambda: torch.randperm(E, device=dev), n=10))
    g = ops.CSRGraph(dst, src, n_dst, n_src)
    X = torch.randn(n_src, 128, device=dev); g.spmm(X)
    m = ops.random_subset_mask(E, 9_000_000, 7, dev)
    print("masked view + its sliced values %.3f ms" % timeit(lambda: g.masked(m).spmm(X), n=10), "(product alone %.3f ms)" % timeit(lambda: g.spmm(X), n=10))
