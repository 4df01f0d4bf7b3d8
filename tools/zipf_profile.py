"""The Zipf(1.2) variant of bench.py (10 M edges, 50 000 destination rows, longest row ~2 M edges) in a loop of its own, for
rocprofv3 --kernel-trace --stats: what the split form (virtual rows + ordered re-sum) spends where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DGMI_SKIP_BUILD", "1")
import torch
from dream_gnn_amd import ops, synth
dev = torch.device("cuda:0")
ND, NS, E, F = 100_000, 50_000, 10_000_000, 128
gen = torch.Generator(device=dev).manual_seed(1)
p = 1.0 / torch.arange(1, NS + 1, device=dev, dtype=torch.float64) ** 1.2
dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=gen).to(torch.int32)
src = torch.randint(0, ND, (E,), generator=gen, device=dev, dtype=torch.int32)
g = ops.CSRGraph(dst, src, NS, ND)
X = torch.randn(ND, F, device=dev)
cj, ci = synth.degree_norm(src, ND), synth.degree_norm(dst, NS)
y = torch.empty(NS, F, device=dev)
deg = (g.indptr[1:] - g.indptr[:-1])
print("rows with < 48 edges: %d of %d, holding %.1f %% of the edges" % (int((deg < 48).sum()), NS, 100.0 * float(deg[deg < 48].sum()) / E))
for _ in range(5):
    g.spmm(X, cj, ci, out=y)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(50):
    g.spmm(X, cj, ci, out=y)
b.record(); torch.cuda.synchronize()
print("zipf(1.2) product: %.4f ms" % (a.elapsed_time(b) / 50))
