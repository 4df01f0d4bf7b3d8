"""(f4) randomized soak of ops.knn_cosine_topk against a float64 brute force: shapes, k, and data
distributions (isotropic, clustered, low-rank, near-duplicates, zero rows, heavy-tailed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(4321 if "--big2" in sys.argv else 1234)


def data(kind, N, D):
    if kind == "iso":
        return torch.randn(N, D, generator=gen, device=dev)
    if kind == "clustered":
        c = torch.randn(max(2, N // 200), D, generator=gen, device=dev)
        idx = torch.randint(0, c.shape[0], (N,), generator=gen, device=dev)
        return c[idx] + 0.05 * torch.randn(N, D, generator=gen, device=dev)
    if kind == "lowrank":
        r = max(2, D // 16)
        return torch.randn(N, r, generator=gen, device=dev) @ torch.randn(r, D, generator=gen, device=dev)
    if kind == "dups":
        x = torch.randn(N, D, generator=gen, device=dev)
        x[N // 2:] = x[: N - N // 2].clone()  # every row has an exact duplicate
        return x
    if kind == "zeros":
        x = torch.randn(N, D, generator=gen, device=dev)
        x[::97] = 0
        return x
    if kind == "heavy":
        return torch.randn(N, D, generator=gen, device=dev) ** 3
    raise ValueError(kind)


bad = 0
cases = 0
# --big: only the sizes of the 256 x 256 screen tiles (N >= 49152: the phase-interleaved LDS-DMA kernel up to k = 16 and 2+ K
# chunks, the register-staged one beyond)
grid = ((55555, 80000, 131072), (96, 384, 768, 1024), (1, 2, 8, 16)) if "--big2" in sys.argv else \
    ((49152, 50007, 66000), (40, 72, 200, 768), (1, 4, 16, 33, 64)) if "--big" in sys.argv else \
    ((1536, 2500, 4097, 9000, 20011, 26000, 45000), (8, 64, 200, 768), (1, 4, 16, 33, 64))
for N in grid[0]:
    for D in grid[1]:
        for k in grid[2]:
            for kind in ("iso", "clustered", "lowrank", "dups", "zeros", "heavy"):
                if N * D > 20011 * 768 and kind not in ("iso", "clustered") and "--big" not in sys.argv and "--big2" not in sys.argv:
                    continue
                if not ops.knn_cosine_supported(N, D, k):
                    continue
                x = data(kind, N, D)
                nrm = x.norm(dim=1, keepdim=True)
                nrm = torch.where(nrm == 0, torch.full_like(nrm, 1e-10), nrm)
                xn = x / nrm
                nbr = ops.knn_cosine_topk(xn, k).long()
                rows = torch.randperm(N, generator=gen, device=dev)[:1500]
                sim = xn[rows].double() @ xn.double().t()
                got = torch.gather(sim, 1, nbr[rows])
                want = torch.topk(sim, k, dim=1).values
                err = float((got - want).abs().max())
                distinct = all(len(set(r.tolist())) == k for r in nbr[rows].cpu())
                valid = int(nbr.min()) >= 0 and int(nbr.max()) < N
                cases += 1
                if err > 3e-6 or not distinct or not valid:
                    bad += 1
                    print("FAIL N=%d D=%d k=%d %s: err %.3g distinct %s valid %s" % (N, D, k, kind, err, distinct, valid), flush=True)
        print("N=%d D=%d done (%d cases, %d bad)" % (N, D, cases, bad), flush=True)
print("soak: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
