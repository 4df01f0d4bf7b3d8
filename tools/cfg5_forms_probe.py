"""Config 5 (node-scaled, 800k x 400k / 80 M edges, 8 ranks): what ONE rank computes per product under
the two sharding forms, on one GPU (no communication):
  A  destination-row shard, X replicated   (shipped): rows 1/8 of N_dst, all sources
  B  source shard, partial Y + reduce-scatter (north_star's literal form): all N_dst rows, 1/8 of the sources
Each with the planned and the XCD-sliced kernel (DGMI_FORCE_KERNEL is not used: layouts are built directly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
F, E = 128, 10_000_000
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=20, warm=3):
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


def case(tag, n_dst, n_src):
    dst = torch.randint(0, n_dst, (E,), generator=g, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    X = torch.randn(n_src, F, device=dev)
    ss, ds = torch.rand(n_src, device=dev), torch.rand(n_dst, device=dev)
    out = torch.empty(n_dst, F, device=dev)
    base = ops.CSRGraph(dst, src, n_dst, n_src)
    t_plan = timeit(lambda: ops.spmm_csr_raw(base.indptr, base.indices, None, X, ss, ds, plan=base.plan, out=out))
    sl = ops.SlicedCSR(dst, src, n_dst, n_src)
    t_sl = timeit(lambda: sl.spmm(X, ss, ds, out=out))
    print("%-64s rows %7d  table %4d MB  avg degree %5.1f : planned %.3f ms  sliced %.3f ms   out block %5.1f MB"
          % (tag, n_dst, n_src * F * 4 // 1_000_000, E / n_dst, t_plan, t_sl, n_dst * F * 4 / 1e6), flush=True)
    del base, sl, X, out


print("form A: destination-row shard (local rows = N_dst / 8, X replicated)")
case("A  drug->disease  (rows: diseases/8, sources: all drugs)", 50_000, 800_000)
case("A  disease->drug  (rows: drugs/8, sources: all diseases)", 100_000, 400_000)
print("form B: source shard (all N_dst rows, sources = N_src / 8; partial Y is reduce-scattered)")
case("B  drug->disease  (rows: all diseases, sources: drugs/8)", 400_000, 100_000)
case("B  disease->drug  (rows: all drugs, sources: diseases/8)", 800_000, 50_000)
print("edge-scaled config 5 (100k x 50k nodes, 80 M edges): form A shard")
case("A  drug->disease  (rows: diseases/8 = 6250, degree 1600)", 6_250, 100_000)
case("A  disease->drug  (rows: drugs/8 = 12500, degree 800)", 12_500, 50_000)
