"""(D3) per-step layout compaction at config-4 size: time per launch series, and — under `rocprofv3 --kernel-trace --stats` —
per kernel.  10 M-edge bipartite layout (unweighted) and the 12.9 M-nnz kNN-64 layout (weighted), 10 % dropped."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__

__graft_entry__.ensure_built()
import torch

from dream_gnn_amd import ops as O, synth

dev = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


drug, dis = synth.bipartite_edges(100_000, 50_000, 10_000_000, seed=0, device=dev)
sl = O.SlicedCSR(dis, drug, 50_000, 100_000)
desc = O.random_subset_select(10_000_000, 9_000_000, 5, dev)
us = timeit(lambda: sl.compacted(desc))
print("bipartite 10 M edges, sliced layout (400 001 pointers), unweighted: %.1f us = %.2f TB/s of the 12 B/edge it must move"
      % (us, 10e6 * 11.6 / us / 1e6), flush=True)
r, c, v = synth.knn_sim_graph(100_000, 64, 21, dev)
sk = O.SlicedCSR(r, c, 100_000, 100_000, vals=v)
dk = O.random_subset_select(int(r.numel()), int(r.numel() * 0.9), 6, dev)
us = timeit(lambda: sk.compacted(dk, sk.vals))
print("kNN-64 %d nnz, sliced layout, weighted: %.1f us = %.2f TB/s of the 19.2 B/edge it must move" % (r.numel(), us, r.numel() * 19.2 / us / 1e6), flush=True)
g = O.CSRGraph(dis, drug, 50_000, 100_000)
from dream_gnn_amd import _lib
us = timeit(lambda: _lib.torch_ops.compact_layout(g.indptr, g.indices, None, g.eid, desc.reshape(1, 8)))
print("bipartite 10 M edges, plain CSR (50 001 pointers): %.1f us" % us, flush=True)
