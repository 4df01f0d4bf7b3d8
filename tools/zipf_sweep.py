import os, sys
sys.path.insert(0, '/root/repo')
import torch
from dream_gnn_amd import ops, synth, _lib
dev = torch.device("cuda:0")
ND, NS, E, F = 100_000, 50_000, 10_000_000, 128
gen = torch.Generator(device=dev).manual_seed(1)
p = 1.0 / torch.arange(1, NS + 1, device=dev, dtype=torch.float64) ** 1.2
dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=gen).to(torch.int32)
src = torch.randint(0, ND, (E,), generator=gen, device=dev, dtype=torch.int32)
X = torch.randn(ND, F, device=dev)
cj, ci = synth.degree_norm(src, ND), synth.degree_norm(dst, NS)
y = torch.empty(NS, F, device=dev)
def t(g):
    for _ in range(5): g.spmm(X, cj, ci, out=y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(30): g.spmm(X, cj, ci, out=y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 30
for light in (48, 0, 24, 96):
    for rows in (512, 256, 192, 128):
        ops.SPLIT_LIGHT_ROW_EDGES = light
        ops.SPLIT_ROW_EDGES = rows
        g = ops.CSRGraph(dst, src, NS, ND)
        ms = t(g)
        print("light < %3d, virtual rows of %4d edges: %d virtual rows  %.4f ms" % (light, rows, g._S.split.n_virtual, ms), flush=True)
        del g
