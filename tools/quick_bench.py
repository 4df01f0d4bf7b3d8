"""Scratch timing of the SpMM kernel on the cfg-4 shapes (not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
F = int(os.environ.get("F", 128))
def run(n_dst, n_src, E, weighted, name):
    g0 = torch.Generator(device="cpu").manual_seed(0)
    dst = torch.randint(0, n_dst, (E,), generator=g0, dtype=torch.int32).to(dev)
    src = torch.randint(0, n_src, (E,), generator=g0, dtype=torch.int32).to(dev)
    vals = torch.rand(E, generator=g0).to(dev) if weighted else None
    t0 = time.time(); g = ops.CSRGraph(dst, src, n_dst, n_src, vals=vals); torch.cuda.synchronize(); t1 = time.time()
    X = torch.randn(n_src, F, device=dev)
    Y = torch.empty(n_dst, F, device=dev)
    for _ in range(5): ops.spmm_csr_raw(g.indptr, g.indices, g.vals, X, out=Y)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in evs:
        a.record(); ops.spmm_csr_raw(g.indptr, g.indices, g.vals, X, out=Y); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs); med = ts[len(ts)//2]
    balg = E*(4*F+4+(4 if weighted else 0)) + n_dst*4*F + (n_dst+1)*4
    print(f"{name}: csr_build {1e3*(t1-t0):.1f} ms | spmm med {med:.3f} ms p10 {ts[3]:.3f} | {E/med/1e6:.2f} Gedge/s | B_alg {balg/1e9:.3f} GB -> {balg/med/1e9:.2f} TB/s = {balg/med/1e9/8*100:.1f}% of 8 TB/s", flush=True)
    # csr build timing
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); ops.csr_from_coo(dst, src, n_dst); b.record(); torch.cuda.synchronize()
    print(f"   csr_from_coo device time {a.elapsed_time(b):.3f} ms")

run(50_000, 100_000, 10_000_000, False, "bip drug->disease")
run(100_000, 50_000, 10_000_000, False, "bip disease->drug")
run(100_000, 100_000, 12_900_000, True, "knn64-like weighted N=100k")
