"""What one rank of an N-rank weak-scaled run computes (no communication): builds rank 0's shards of
the N x config-4 problem on this single GPU and times the 8 local products."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda:0")
scale = os.environ.get("SCALE", "edges")
for world in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    ops, build_ms, (nd, ns, E, knn_k) = bench.build_ops(torch, 0, world, dev, scale)
    torch.cuda.synchronize(); setup = time.perf_counter() - t0
    for _ in range(3):
        for op in ops: op.launch(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        for op in ops: op.launch(True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    edges = sum(op.nnz for op in ops)
    per = {op.name: round(sum(a.elapsed_time(b) for a, b in op.events) / len(op.events), 3) for op in ops}
    gather_mb = sum(op.y_local.numel() * 4 for op in ops) / 1e6
    print(f"world {world} [{scale}-scaled]: global {nd}x{ns} E={E} kNN-{knn_k}; rank-0 local edges/step {edges/1e6:.1f}M; compute {dt*1e3:.2f} ms/step -> {edges/dt/1e9:.1f} Gedge/s per rank; "
          f"all-gather payload contributed {gather_mb:.0f} MB/step; setup {setup:.1f} s; peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
    print("   ", per, flush=True)
    del ops
