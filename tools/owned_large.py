"""Row-owned kernel vs planned / sliced on feature tables too large for the XCD-sliced kernel
(10 M edges, F = 128, 50k destination rows): the node-scaled config-5 shard regime."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops
from tools.owned_bench import timeit

dev = torch.device("cuda:0")
F, E, n_dst = 128, 10_000_000, 50_000
g = torch.Generator(device=dev).manual_seed(0)
for n_src in (150_000, 200_000, 400_000, 800_000):
    dst = torch.randint(0, n_dst, (E,), generator=g, device=dev, dtype=torch.int32)
    src = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    X = torch.randn(n_src, F, device=dev)
    ss, ds = torch.rand(n_src, device=dev), torch.rand(n_dst, device=dev)
    base = ops.CSRGraph(dst, src, n_dst, n_src)
    y_ref = ops.spmm_csr_raw(base.indptr, base.indices, None, X, ss, ds, plan=base.plan)
    out = torch.empty_like(y_ref)
    t_plan = timeit(lambda: ops.spmm_csr_raw(base.indptr, base.indices, None, X, ss, ds, plan=base.plan, out=out))
    sl = ops.SlicedCSR(dst, src, n_dst, n_src)
    t_sl = timeit(lambda: sl.spmm(X, ss, ds, out=out))
    line = "table %4d MB: planned %.4f  sliced %.4f " % (n_src * F * 4 // 1_000_000, t_plan, t_sl)
    for m in (5, 4):
        for skb in (3200, 6400, 12800):
            os.environ["DGMI_OWNED_SLICE_KB"] = str(skb)
            ow = ops.OwnedCSR(dst, src, n_dst, n_src, F=F, blocks_per_cu=m, paced=False)
            t = timeit(lambda: ow.spmm(X, ss, ds, out=out))
            assert float((out - y_ref).abs().max() / y_ref.abs().max()) < 1e-5
            line += " owned(m%d,S%d) %.4f" % (m, ow.geom.n_slices, t)
            del ow
    print(line, flush=True)
