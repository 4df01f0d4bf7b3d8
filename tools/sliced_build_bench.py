"""(f1) sliced layout: sorted from the COO list vs derived from the CSR (one partition pass), 10 M edges."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for n_dst, n_src, E in ((50_000, 100_000, 10_000_000), (100_000, 50_000, 10_000_000), (100_000, 100_000, 12_900_000)):
    dst = torch.randint(0, n_dst, (E,), generator=g, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    indptr, indices, eid, _ = ops.csr_from_coo(dst, src, n_dst, n_src, return_flag=True)

    def timeit(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    t_csr = timeit(lambda: ops.csr_from_coo(dst, src, n_dst, n_src, return_flag=True))
    t_coo = timeit(lambda: ops.SlicedCSR(dst, src, n_dst, n_src))
    t_from = timeit(lambda: ops.SlicedCSR.from_csr(indptr, indices, eid, n_dst, n_src))
    print("%d x %d, %d edges: CSR %.3f ms; sliced from COO %.3f ms; sliced from the CSR %.3f ms" % (n_dst, n_src, E, t_csr, t_coo, t_from), flush=True)
