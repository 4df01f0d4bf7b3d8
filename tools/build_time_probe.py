"""How long one rank of `bench.py --gpus 8` spends before its first timed step (node-scaled and edge-scaled
workloads), on one GPU, no communication: data generation + graph builds + first products."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for scale in ("nodes", "edges"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops, build_ms, shape = bench.build_ops(torch, 0, world, dev, scale)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        for op in ops:
            op.launch(False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("world %d, %s-scaled %s: build_ops %.1f s (graph-side %.0f ms), 3 steps %.1f ms, peak memory %.1f GB"
          % (world, scale, shape, t1 - t0, sum(build_ms.values()), (t2 - t1) * 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del ops
    torch.cuda.empty_cache()
