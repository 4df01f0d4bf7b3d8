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
    # (row sums as differences of a running sum between the row boundaries: exact for these integer values and without the
    # float64 atomics of index_add_ — graph._normalized_adjacency)
    bounds = torch.searchsorted(idx[0].contiguous(), torch.arange(n + 1, device=device))
    run = torch.cat([torch.zeros(1, dtype=torch.float64, device=device), val.cumsum(0)])
    rowsum = run[bounds[1:]] - run[bounds[:-1]]
    val = (val / rowsum[idx[0]]).to(torch.float32)
    return idx[0].to(torch.int32), idx[1].to(torch.int32), val


def degree_norm(ids: torch.Tensor, n: int) -> torch.Tensor:
    """``1/sqrt(degree)`` with 0 for isolated nodes — the ci/cj of data_loader.py:454-457."""
    deg = torch.bincount(ids.long(), minlength=n).to(torch.float32)
    return torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))


# ---------------------------------------------------------------------------------------------
# dataset-SHAPED problems for BASELINE configs 2 / 3 (the .mat files are absent: SURVEY.md §8c)
# ---------------------------------------------------------------------------------------------
#: (n_drug, n_disease, known associations) per dataset — sizes from the literature,
#: not from the reference, whose data files are missing.
DATASET_SHAPES = {"lrssl": (763, 681, 3051), "Cdataset": (663, 409, 2532), "Gdataset": (593, 313, 1933)}


def dataset_shaped_pairs(blocks, train_frac: float = 0.9, seed: int = 0):
    """Train pairs of one cross-validation fold in the reference's format — data_loader.py:136-203:
    ALL drug x disease cells are samples (positives and negatives, no down-sampling, :146-150,170),
    split by label, ``train_frac`` of each kept.  ``blocks`` = [(n_drug, n_dis, n_pos), ...]: several
    blocks give the block-diagonal union SURVEY.md §8(d) assumes for "Cdataset + Gdataset merged"
    (no pair crosses datasets).  Returns CPU tensors ``(drug_ids, dis_ids, labels, n_drug, n_dis)``,
    deterministic in ``seed`` on any box (numpy generator)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    drugs, diss, labels = [], [], []
    d0 = s0 = 0
    for nd, ns, n_pos in blocks:
        cells = nd * ns
        pos = rng.choice(cells, n_pos, replace=False)
        is_pos = np.zeros(cells, bool)
        is_pos[pos] = True
        for lab in (1, 0):
            ids = np.flatnonzero(is_pos == bool(lab))
            ids = rng.permutation(ids)[: int(round(train_frac * ids.size))]
            drugs.append(ids // ns + d0)
            diss.append(ids % ns + s0)
            labels.append(np.full(ids.size, lab, np.float32))
        d0, s0 = d0 + nd, s0 + ns
    order = rng.permutation(sum(a.size for a in drugs))  # KFold shuffles the sample order
    cat = lambda parts, dt: torch.from_numpy(np.concatenate(parts)[order].astype(dt))
    return cat(drugs, np.int64), cat(diss, np.int64), cat(labels, np.float32), d0, s0


def dataset_shaped_batch(blocks, emb: int = 768, k: int = 4, seed: int = 0, device="cpu"):
    """Everything ``Net.forward`` consumes for one fold of a dataset-shaped problem: encoder /
    decoder graphs over the train pairs, the two kNN-``k`` similarity graphs and the two feature
    kNN graphs (data_loader.py:278-344, through ``graph.similarity_graph`` /
    ``feature_similarity_graph``), similarity rows as FGCN features (train.py:174-175) and
    L2-normalised ``emb``-wide embeddings (data_loader.py:221-222).  Generated on the CPU (same
    bits on every box), then moved to ``device``."""
    from . import graph as G

    drug, dis, labels, nd, ns = dataset_shaped_pairs(blocks, seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    batch = {}
    for key, n in (("drug", nd), ("disease", ns)):
        sim = torch.rand(n, n, generator=gen)
        sim = (sim + sim.t()) / 2
        sim.fill_diagonal_(1.0)
        feat = torch.nn.functional.normalize(torch.randn(n, emb, generator=gen))
        batch[key + "_sim_feat"], batch[key + "_feat"] = sim, feat
        batch["_%s_sim_graph_coo" % key] = _cpu_knn(sim, k)
        xn = feat / feat.norm(dim=1, keepdim=True)
        batch["_%s_feat_graph_coo" % key] = _cpu_knn(xn @ xn.t(), k)
    out = {"enc_pairs": (drug, dis, labels), "n_drug": nd, "n_dis": ns}
    dev = torch.device(device)
    for key, n in (("drug", nd), ("disease", ns)):
        out[key + "_sim_feat"] = batch[key + "_sim_feat"].to(dev)
        out[key + "_feat"] = batch[key + "_feat"].to(dev)
        for name, src in ((key + "_graph", "_%s_sim_graph_coo" % key), (key + "_feature_graph", "_%s_feat_graph_coo" % key)):
            r, c, v = batch[src]
            out[name] = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev)
    out["enc_graph"] = G.build_enc_graph(drug, dis, labels, nd, ns, device=dev).int()
    out["dec_graph"] = G.build_dec_graph(drug, dis, nd, ns, device=dev).int()
    return out, labels.to(dev)


def _cpu_knn(sim: torch.Tensor, k: int):
    """data_loader.py:278-310 on the CPU with plain torch: top-k -> A + A^T -> + I -> D^-1 (.);
    entries row-major sorted.  (row, col, val) int64 / int64 / fp32."""
    n = sim.shape[0]
    nbr = torch.topk(sim, min(k, n - 1), dim=1).indices
    rows = torch.arange(n).repeat_interleave(nbr.shape[1])
    eye = torch.arange(n)
    r = torch.cat([rows, nbr.reshape(-1), eye])
    c = torch.cat([nbr.reshape(-1), rows, eye])
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), torch.ones(r.numel(), dtype=torch.float64), (n, n)).coalesce()
    idx, val = adj.indices(), adj.values()
    rowsum = torch.zeros(n, dtype=torch.float64).index_add_(0, idx[0], val)
    return idx[0], idx[1], (val / rowsum[idx[0]]).to(torch.float32)


def net_args(emb: int = 768, out_units: int = 128, layers: int = 3, agg_units: int = 1024, nhid1: int = 768,
             n_drug: int = 763, n_dis: int = 681, dropout: float = 0.3, attention_dropout: float = 0.5, device=None):
    """The reference's hyper-parameters as ``train.py:404-448`` defaults them (``--gcn_agg_units
    1024`` -> layer-0 width 341, ``--gcn_out_units 128``, ``--layers 3``, ``--nhid1 768``; config 3
    passes 256 for the two output widths)."""
    import types

    return types.SimpleNamespace(rating_vals=[0, 1], src_in_units=emb, dst_in_units=emb, gcn_agg_units=agg_units,
                                 gcn_out_units=out_units, dropout=dropout, gcn_agg_accum="sum",
                                 model_activation="leaky", share_param=True, device=device, layers=layers,
                                 fdim_drug=n_drug, fdim_disease=n_dis, nhid1=nhid1, nhid2=out_units,
                                 attention_dropout=attention_dropout)
