"""One config-4 product per kernel form, a few launches each, for rocprofv3 PMC passes:
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d DIR -- python3 tools/owned_prof.py
Forms are told apart in the trace by kernel name + launch order (printed here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")
F, ND, NS, E = 128, 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(ND, NS, E, seed=0, device=dev)
cj, ci = synth.degree_norm(drug, ND), synth.degree_norm(dis, NS)
xd = torch.randn(ND, F, device=dev)
out = torch.empty(NS, F, device=dev)
forms = os.environ.get("FORMS", "sliced,4:16,4:32,5:32,5:16").split(",")
for f in forms:
    if f == "sliced":
        k = ops.SlicedCSR(dis, drug, NS, ND)
    else:
        m, s = (int(v) for v in f.split(":"))
        k = ops.OwnedCSR(dis, drug, NS, ND, F=F, blocks_per_cu=m, n_slices=s)
    for _ in range(4):
        k.spmm(xd, cj, ci, out=out)
    torch.cuda.synchronize()
    print("form", f, flush=True)
