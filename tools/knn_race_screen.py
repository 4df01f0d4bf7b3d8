"""(f4) race screen of the LDS-DMA screen kernel: the same search repeated while a second stream keeps the memory system and
the CUs busy with other work (timing of the DMA landings and of the waves changes from run to run); every run must return the
same neighbour ids — the rescoring orders by (score desc, id asc), so the answer does not depend on the order of the appends."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(99)
bad = 0
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.uint8)
for N, D, k in ((60000, 768, 4), (50001, 200, 16), (70000, 384, 64), (131072, 96, 8)):
    x = torch.randn(N, D, generator=gen, device=dev)
    x[1000:1200] = x[1000] + 1e-3 * torch.randn(200, D, generator=gen, device=dev)
    xn = x / x.norm(dim=1, keepdim=True)
    ref = ops.knn_cosine_topk(xn, k).clone()
    torch.cuda.synchronize()
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(3 + rep % 4):
                if rep % 3 == 0:
                    _ = a @ a          # MFMA + L2 pressure
                elif rep % 3 == 1:
                    big.add_(1)        # HBM streaming
                else:
                    _ = a.float().sum(dim=0)
        got = ops.knn_cosine_topk(xn, k)
        torch.cuda.synchronize()
        if not torch.equal(got, ref):
            bad += 1
            print("MISMATCH N=%d D=%d k=%d rep %d: %d rows differ" % (N, D, k, rep, int((got != ref).any(dim=1).sum())), flush=True)
    print("N=%d D=%d k=%d: 12 repeats under load done (%d bad so far)" % (N, D, k, bad), flush=True)
print("race screen: %d mismatching runs" % bad)
sys.exit(1 if bad else 0)
