"""Device-side generators for the synthetic configs of BASELINE.json (cfg 4 / cfg 5).

Setup code only (never inside a timed region): plain torch ops on the GPU.  The shapes follow
SURVEY.md §8(d): a uniform bipartite drug x disease graph with exactly E distinct cells, and
kNN-style similarity graphs pushed through the reference's own adjacency pipeline
(data_loader.py:297-308 + utils.py:11-27): ones COO -> A + A^T -> + I -> D^-1 (.) -> COO fp32.
"""
from __future__ import annotations

import torch


def bipartite_edges(n_drug: int, n_dis: int, E: int, seed: int, device):
    """E distinct (drug, disease) cells, uniform, in random order -> (drug_ids, dis_ids) int32."""
    cells = n_drug * n_dis
    if E > cells:
        raise ValueError("more edges than cells")
    g = torch.Generator(device=device).manual_seed(seed)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = E
    while keys.numel() < E:  # top-up until E distinct cells exist (dedup removes ~E^2/2cells)
        extra = torch.randint(0, cells, (int(want * 1.02) + 1024,), generator=g, device=device)
        keys = torch.unique(torch.cat([keys, extra]))
        want = E - keys.numel() if keys.numel() < E else 0
    perm = torch.randperm(keys.numel(), generator=g, device=device)
    keys = keys[perm[:E]]
    return (keys // n_dis).to(torch.int32), (keys % n_dis).to(torch.int32)


def knn_sim_graph(n: int, k: int, seed: int, device):
    """Row-normalised symmetrised kNN adjacency with self loops as COO (row, col, val).

    Neighbours are k random ids per row (the reference's dense-similarity argpartition,
    data_loader.py:293, needs an n x n matrix — 40 GB at n = 100k); everything after the
    neighbour choice is the reference's pipeline.  Entries come out row-major sorted, values
    fp32, like ``sparse_mx_to_torch_sparse_tensor`` (utils.py:20-27).  nnz ~= (2k+1) n.
    """
    g = torch.Generator(device=device).manual_seed(seed)
    rows = torch.arange(n, device=device).repeat_interleave(k)
    cols = torch.randint(0, n, (n * k,), generator=g, device=device)
    eye = torch.arange(n, device=device)
    r = torch.cat([rows, cols, eye])
    c = torch.cat([cols, rows, eye])
    v = torch.ones(r.numel(), dtype=torch.float64, device=device)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).coalesce()  # A + A^T + I
    idx, val = adj.indices(), adj.values()
    rowsum = torch.zeros(n, dtype=torch.float64, device=device).index_add_(0, idx[0], val)
    val = (val / rowsum[idx[0]]).to(torch.float32)
    return idx[0].to(torch.int32), idx[1].to(torch.int32), val


def degree_norm(ids: torch.Tensor, n: int) -> torch.Tensor:
    """``1/sqrt(degree)`` with 0 for isolated nodes — the ci/cj of data_loader.py:454-457."""
    deg = torch.bincount(ids.long(), minlength=n).to(torch.float32)
    return torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))
