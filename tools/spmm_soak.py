"""Randomized soak of ops.CSRGraph.spmm / spmm_t (every kernel form the size rules pick, with and without
values, scalings and on-the-fly edge dropout) against float64 torch.sparse on the same GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(99)
cpu_gen = torch.Generator().manual_seed(7)


def ref(dst, src, vals, n_dst, n_src, X, ss, ds, transpose=False):
    v = torch.ones(dst.shape[0], dtype=torch.float64, device=dev) if vals is None else vals.double()
    A = torch.sparse_coo_tensor(torch.stack([dst.long(), src.long()]), v, (n_dst, n_src))
    if transpose:
        A = A.t()
        ss, ds = ds, ss
    Xd = X.double()
    if ss is not None:
        Xd = Xd * ss.double()[:, None]
    Y = torch.sparse.mm(A.coalesce(), Xd)
    if ds is not None:
        Y = Y * ds.double()[:, None]
    return Y


shapes = [  # (n_dst, n_src, E, F): around the rules' thresholds (sliced: 10-160 MB tables at degree >= 64; column
    # passes: slice > 4 MiB and >= 32768 rows; split: irregular degrees)
    (300, 200, 20_000, 128), (5000, 3000, 400_000, 344), (33_000, 70_000, 2_400_000, 128), (32_000, 70_000, 2_300_000, 128),
    (40_000, 60_000, 2_800_000, 128), (40_000, 66_000, 2_800_000, 256), (20_000, 120_000, 1_500_000, 64),
    (50_000, 100_000, 3_500_000, 128), (763, 681, 465_000, 344), (1000, 900, 30_000, 341), (70_000, 30_000, 5_000_000, 32)]
bad = cases = 0
for n_dst, n_src, E, F in shapes:
    for skew in (False, True):
        if skew:
            p = 1.0 / torch.arange(1, n_dst + 1, dtype=torch.float64) ** 1.1
            dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=cpu_gen).to(torch.int32).to(dev)
        else:
            dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        for weighted in (False, True, "row scale x multiplicity", "column scale x multiplicity"):
            if weighted is True:
                vals = torch.randn(E, generator=gen, device=dev)
            elif weighted:  # (r4) the reference's adjacency format and its transpose: the value stream is dropped (ops._mult_form)
                scale = torch.rand(n_dst if weighted.startswith("row") else n_src, generator=gen, device=dev) + 0.05
                mult = torch.randint(1, 9, (E,), generator=gen, device=dev).float()
                vals = mult * scale[(dst if weighted.startswith("row") else src).long()]
            else:
                vals = None
            g = ops.CSRGraph(dst, src, n_dst, n_src, vals=vals)
            X = torch.randn(n_src, F, generator=gen, device=dev)
            dY = torch.randn(n_dst, F, generator=gen, device=dev)
            ss = torch.rand(n_src, generator=gen, device=dev) + 0.1
            ds = torch.rand(n_dst, generator=gen, device=dev) + 0.1
            for scaled in (False, True):
                a, b = (ss, ds) if scaled else (None, None)
                y = g.spmm(X, a, b)
                yr = ref(dst, src, vals, n_dst, n_src, X, a, b)
                e1 = float((y.double() - yr).abs().max() / yr.abs().max().clamp_min(1e-30))
                dx = g.spmm_t(dY, a, b)
                dr = ref(dst, src, vals, n_dst, n_src, dY, a, b, transpose=True)
                e2 = float((dx.double() - dr).abs().max() / dr.abs().max().clamp_min(1e-30))
                # on-the-fly dropout against the product over the kept edges only
                keep = int(E * 0.9)
                desc = ops.random_subset_select(E, keep, 4242 + cases, dev)
                m = ops.keep_mask(desc, E).bool()
                ops.COMPACT_DROPPED = bool(cases % 2)  # (r4) alternate: compacted layouts / on-the-fly KEEP kernels
                view = g.dropped(desc)
                yd = view.spmm(X, a, b)
                ydr = ref(dst[m], src[m], None if vals is None else vals[m], n_dst, n_src, X, a, b)
                e3 = float((yd.double() - ydr).abs().max() / ydr.abs().max().clamp_min(1e-30))
                dxd = view.spmm_t(dY, a, b)
                dxr = ref(dst[m], src[m], None if vals is None else vals[m], n_dst, n_src, dY, a, b, transpose=True)
                e3 = max(e3, float((dxd.double() - dxr).abs().max() / dxr.abs().max().clamp_min(1e-30)))
                ops.COMPACT_DROPPED = True
                cases += 4
                if max(e1, e2, e3) > 2e-5 or int(m.sum()) != keep:
                    bad += 1
                    print("FAIL %s skew=%s weighted=%s scaled=%s: fwd %.2g bwd %.2g dropped %.2g kept %d" %
                          ((n_dst, n_src, E, F), skew, weighted, scaled, e1, e2, e3, int(m.sum())), flush=True)
            del g
    print("%s done (%d products, %d bad)" % ((n_dst, n_src, E, F), cases, bad), flush=True)
print("soak: %d products, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
