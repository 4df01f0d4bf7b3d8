"""Zipf(1.2) destination degrees, 10 M edges on the 100k-source table: virtual-row length of ops._SplitSliced
x column passes of the sliced kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth
import bench

dev = torch.device("cuda:0")
F, E, n_dst, n_src = 128, 10_000_000, 50_000, 100_000
X = torch.randn(n_src, F, device=dev)
out = torch.empty(n_dst, F, device=dev)
ss, ds = torch.rand(n_src, device=dev), torch.rand(n_dst, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)  # bench.py's Zipf variant
p = 1.0 / torch.arange(1, n_dst + 1, device=dev, dtype=torch.float64) ** 1.2
gz_dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=gen).to(torch.int32)
gz_src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
for split in (2048, 1024, 512, 256):
    for full in (True, False):
        ops.SPLIT_ROW_EDGES = split
        gz = ops.CSRGraph(gz_dst, gz_src, n_dst, n_src)
        orig = ops.SlicedCSR.spmm
        if not full:  # let the launcher choose the column passes by footprint
            def patched(self, *a, full_width=False, **k):
                return orig(self, *a, full_width=False, **k)
            ops.SlicedCSR.spmm = patched
        try:
            t = bench.timeit(torch, lambda: gz.spmm(X, ss, ds, out=out), reps=10, warm=3)
        finally:
            ops.SlicedCSR.spmm = orig
        print("virtual rows <= %4d edges, %s: %.4f ms" % (split, "one full-width pass" if full else "column passes by footprint", t), flush=True)
        del gz
