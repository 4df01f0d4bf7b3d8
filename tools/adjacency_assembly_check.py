"""(D2) `graph._coalesce_unit_entries` against `torch.sparse_coo_tensor(...).coalesce()`: identical indices and values, and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import graph as G

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(3)
ok = True
for n, k in ((1000, 4), (763, 7), (50000, 64), (100000, 64), (100000, 4), (5, 2)):
    nbr = torch.randint(0, n, (n, k), generator=gen, device=dev)
    rows = torch.arange(n, device=dev).repeat_interleave(k)
    cols = nbr.reshape(-1)
    eye = torch.arange(n, device=dev)
    r, c = torch.cat([rows, cols, eye]), torch.cat([cols, rows, eye])
    def lib():
        a = torch.sparse_coo_tensor(torch.stack([r, c]), torch.ones(r.numel(), dtype=torch.float64, device=dev), (n, n)).coalesce()
        return a.indices(), a.values()
    def own():
        return G._coalesce_unit_entries(r, c, n)
    res = []
    for fn in (lib, own):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            out = fn()
        b.record(); torch.cuda.synchronize()
        res.append((a.elapsed_time(b) / 3, out))
    same = torch.equal(res[0][1][0], res[1][1][0]) and torch.equal(res[0][1][1], res[1][1][1])
    ok &= same
    print("n=%d k=%d: coalesce %.2f ms, own %.2f ms, identical %s (nnz %d)" % (n, k, res[0][0], res[1][0], same, res[1][1][1].numel()), flush=True)
sys.exit(0 if ok else 1)
