"""What the memory hierarchy gives whole-row gathers in the SpMM kernels' access shape
(`dgmi_probe_row_gather_f32`), per cache level, next to the gather kernels on an L2-resident table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops, _lib

dev = torch.device("cuda:0")
F, E = 128, 10_000_000


def timeit(fn, reps=30, warm=5):
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


def probe(table_rows, window, per_xcd, groups=1024 * 8 * 5, per_group=1024):
    T = torch.randn(table_rows, F, device=dev)
    out = torch.empty(groups, F, device=dev)
    fn = lambda: _lib.check(_lib.lib.dgmi_probe_row_gather_f32(T.data_ptr(), table_rows, F, groups, per_group, window,
                                                              per_xcd, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "probe")
    return groups * per_group * F * 4 / timeit(fn, reps=10, warm=2) / 1e9  # TB/s


for rows, win, px, tag in ((4096, 4096, 0, "2 MB table, shared by all (L2)"), (8 * 6250, 6250, 1, "8 x 3.2 MB windows, one per XCD (L2-local)"),
                           (8 * 12500, 12500, 1, "8 x 6.4 MB windows, one per XCD"), (100_000, 100_000, 0, "51 MB table, uniform (Infinity Cache)"),
                           (800_000, 800_000, 0, "410 MB table, uniform (HBM + IC)"), (3_200_000, 3_200_000, 0, "1.6 GB table, uniform (HBM)")):
    print("probe %-50s %.2f TB/s" % (tag, probe(rows, win, px)), flush=True)

g = torch.Generator(device=dev).manual_seed(0)
for n_dst, n_src in ((50_000, 4096), (100_000, 4096)):
    dst = torch.randint(0, n_dst, (E,), generator=g, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    X = torch.randn(n_src, F, device=dev)
    ss, ds = torch.rand(n_src, device=dev), torch.rand(n_dst, device=dev)
    out = torch.empty(n_dst, F, device=dev)
    base = ops.CSRGraph(dst, src, n_dst, n_src)
    alg = E * (4 * F + 4) / 1e9
    t = timeit(lambda: ops.spmm_csr_raw(base.indptr, base.indices, None, X, ss, ds, plan=base.plan, out=out))
    print("n_dst %d, 2 MB table: planned %.4f ms (%.1f TB/s)" % (n_dst, t, alg / t), flush=True)
    sl = ops.SlicedCSR(dst, src, n_dst, n_src)
    t = timeit(lambda: sl.spmm(X, ss, ds, out=out))
    print("   xcd-sliced pair %.4f ms (%.1f TB/s)" % (t, alg / t), flush=True)
