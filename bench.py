#!/usr/bin/env python3
"""bench.py — GCMC+FGCN SpMM edges/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts its N ranks itself
(`self_launch`: N fresh child processes of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, spawned
before this process has made any GPU call; rank 0's JSON line is forwarded as the last line of stdout; the first
rank that fails, or the time limit, ends the others).

Workload (N = 1): BASELINE config 4 — synthetic bipartite 100k drugs x 50k diseases with
10M edges plus kNN-64 similarity graphs on both node sets, F = 128, fp32.  One *step* is one
pass of the message-passing hot path over those graphs, forward and backward:

    GCMC   copy_u->sum with the cj/ci scalings fused   (reference layers.py:224-234)
             drug->disease, disease->drug, and the transpose (autograd) of each      4 launches
    FGCN   th.spmm(adj, support)                       (reference layers.py:312)
             drug-kNN, disease-kNN, and the transpose of each                         4 launches

`value` = edges processed per second over the whole job, inputs resident in HBM.

N > 1 (BASELINE config 5, weak scaling): SURVEY.md §8(d) fixes config 5 as N x the nodes AND N x
the edges of config 4 (800k x 400k / 80 M edges at N = 8) — that is the primary workload; the
other reading (config 4's node set with N x the edges) is measured in the same run with equal
standing and reported under `edge_scaled` (`--scale edges` swaps the two).  Every rank owns a
contiguous block of destination rows of every graph — cut by nnz (`balanced_row_bounds`) — and all
their in-edges; each local SpMM is followed by the exchange of its row block over RCCL
(dream_gnn_amd/shard.py: one RCCL all-gather per product — the default, and the only form the judged reading
uses unless asked; `DGMI_EXCHANGE=direct` selects the all-links batched point-to-point form, `=auto` times both at
start-up, guarded, and keeps the faster: both opt-in, the point-to-point form has never run on real links),
overlapped with the next SpMM; time = max over ranks.

The JSON line carries, besides the contract fields:
  roofline      dominant product (GCMC, unweighted, scaled, F=128): `achieved` = §8(d) ALGORITHMIC
                bytes / HIP-event time (the contract's definition — re-reads served by L2 / Infinity
                Cache count, so `frac` may exceed 1), `traffic` = fabric bytes per launch from the
                committed rocprofv3 PMC passes, and bounded readings beside it: `hbm_traffic_frac`
                (measured bytes / time / HBM peak), `l2_gather` (the same product against the
                row-gather rate a probe kernel reaches in this process, in the kernel's own access
                shape and L2 regime), `compulsory_bytes`.
  kernels       per product: time, edges/s, algorithmic and compulsory bytes, both fractions.
  variants      config-4-sized Zipf(1.2) product, the edge-dropped products the training step runs
                (train.py:267), config 2 / 3 products as us per call.
  cpu_baseline  the OpenMP CPU oracle timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_COPY_GBS = 6290.0  # measured float4 copy, same table
F = 128
BASE_DRUG, BASE_DIS, BASE_EDGES, KNN_K = 100_000, 50_000, 10_000_000, 64
# The GCMC product at F=128 on config 4's 25-51 MB feature tables runs as the XCD-local pair behind a row-scale pass:
# scale_rows (diag(cj) X, one streaming pass) + gather kernel (unweighted) + the 8-plane reduce (dst scale).  Its time
# is taken with HIP events around the three; rocprofv3's averages of the kernels add up to it.
# 50k-source direction: 32-lane groups; 100k-source direction: 16-lane groups, two column passes (dgmi_sliced.hip)
DOMINANT = "scale_rows_kernel<4> + spmm_sliced_vec4_kernel<{32|16},0,false,false,true> + reduce_planes_kernel<true,8>"


def algorithmic_bytes(nnz, n_rows, weighted, n_scales_src=0, n_scales_dst=0, width=F):
    """nnz*(4F + 4 + 4w) + N_dst*4F + (N_dst+1)*4 (+ fused scale vectors)."""
    return nnz * (4 * width + 4 + (4 if weighted else 0)) + n_rows * 4 * width + (n_rows + 1) * 4 \
        + 4 * n_scales_src + 4 * n_scales_dst


def compulsory_bytes(nnz, n_rows, n_src, weighted, n_scales_src=0, n_scales_dst=0, width=F):
    """every index (and value) once, every source row once, Y once."""
    return nnz * (4 + (4 if weighted else 0)) + (n_src + n_rows) * 4 * width + (n_rows + 1) * 4 \
        + 4 * n_scales_src + 4 * n_scales_dst


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr (rank 0 / the single process): a long N > 1 run — graph generation for 8 x config 4, two
    workloads — must not look hung to whoever watches its output."""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("[bench.py %7.1f s] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


def timeit(torch, fn, reps=30, warm=5):
    """Average ms per call by HIP events on the current stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


class Op:
    """One SpMM of the step: this rank's row block of Y = diag(ds) A diag(ss) X (+ exchange)."""

    def __init__(self, torch, name, shard, X, ss, ds, weighted, dominant):
        self.torch = torch
        self.name, self.shard, self.X, self.ss, self.ds = name, shard, X, ss, ds
        self.weighted, self.dominant = weighted, dominant
        rows = shard.hi - shard.lo
        self.y_local = torch.empty((rows, F), dtype=torch.float32, device=X.device)
        self.y_full = None if shard.world == 1 else torch.empty((shard.n_dst, F), dtype=torch.float32, device=X.device)
        self.nnz = shard.nnz
        n_ss, n_ds = (0 if ss is None else ss.numel()), (0 if ds is None else rows)
        self.bytes = algorithmic_bytes(self.nnz, rows, weighted, n_ss, n_ds)
        self.compulsory = compulsory_bytes(self.nnz, rows, shard.n_src, weighted, n_ss, n_ds)
        self.table_bytes = shard.n_src * F * 4
        self.events = []

    def launch(self, record):
        if record:
            a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            a.record()
        self.shard.spmm_local(self.X, self.ss, self.ds, out=self.y_local)
        if record:
            b.record()
            self.events.append((a, b))


def build_ops(torch, rank, world, dev, scale):
    """The 8 products of a step on this rank.  Returns (ops, build_ms, shape): build_ms is the time
    of the graph-side work only (row masks, device COO->CSR, launch plans, validation readback),
    measured around each shard's construction after its edge list exists; the synthetic data
    generation is not part of it."""
    from dream_gnn_amd import shard as S
    from dream_gnn_amd import synth

    if scale == "nodes":
        nd, ns, E, knn_k = BASE_DRUG * world, BASE_DIS * world, BASE_EDGES * world, KNN_K
    else:
        nd, ns, E, knn_k = BASE_DRUG, BASE_DIS, BASE_EDGES * world, KNN_K * world
    drug, dis = synth.bipartite_edges(nd, ns, E, seed=0, device=dev)          # identical on every rank
    cj_drug, ci_dis = synth.degree_norm(drug, nd), synth.degree_norm(dis, ns)  # symm: ci == cj per type
    g = torch.Generator(device=dev).manual_seed(3)
    x_drug = torch.randn((nd, F), generator=g, device=dev)
    x_dis = torch.randn((ns, F), generator=g, device=dev)
    build_ms = {}
    warm = torch.arange(4096, dtype=torch.int32, device=dev)
    S.RowShard(warm % 64, warm % 128, 64, 128, torch.tensor([0, 64]), 0).local.spmm(torch.ones(128, F, device=dev))
    del warm  # first-call costs (code-object load, rocPRIM temp sizing) are not graph-build time

    def make(name, dst, src, n_dst, n_src, vals=None):
        # rows cut by nnz: each rank holds ~1/world of the edges, whatever the degrees
        # (equal row counts when those already balance the edges within 1 %: unpadded exchange)
        deg = torch.bincount(dst.long(), minlength=n_dst)
        bounds = S.choose_row_bounds(deg, world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh = S.RowShard(dst, src, n_dst, n_src, bounds, rank, vals=vals)
        torch.cuda.synchronize()
        build_ms[name] = (time.perf_counter() - t0) * 1e3
        return sh

    ops = []
    # GCMC: drug -> disease (rows = diseases) and disease -> drug (rows = drugs); backward = reversed edges
    ops.append(Op(torch, "gcmc_fwd drug->disease", make("gcmc_fwd drug->disease", dis, drug, ns, nd), x_drug, cj_drug, ci_dis, False, True))
    ops.append(Op(torch, "gcmc_fwd disease->drug", make("gcmc_fwd disease->drug", drug, dis, nd, ns), x_dis, ci_dis, cj_drug, False, True))
    # dX = diag(cj) A^T diag(ci) dY: rows = sources; the reversed edge list partitioned by source range
    ops.append(Op(torch, "gcmc_bwd drug->disease", make("gcmc_bwd drug->disease", drug, dis, nd, ns), x_dis, ci_dis, cj_drug, False, True))
    ops.append(Op(torch, "gcmc_bwd disease->drug", make("gcmc_bwd disease->drug", dis, drug, ns, nd), x_drug, cj_drug, ci_dis, False, True))
    del drug, dis
    # FGCN: row-normalised symmetrised kNN-64 graphs (values are not symmetric -> real transpose for bwd)
    for tag, n, x, seed in (("drug", nd, x_drug, 21), ("disease", ns, x_dis, 22)):
        r, c, v = synth.knn_sim_graph(n, knn_k, seed, dev)
        ops.append(Op(torch, "fgcn_fwd %s-knn" % tag, make("fgcn_fwd %s-knn" % tag, r, c, n, n, v), x, None, None, True, False))
        ops.append(Op(torch, "fgcn_bwd %s-knn" % tag, make("fgcn_bwd %s-knn" % tag, c, r, n, n, v), x, None, None, True, False))
        del r, c, v
    t0 = time.perf_counter()
    for op in ops:  # lazily built layouts (sliced CSR) exist before anything is timed, whatever --warmup says
        op.launch(False)
    torch.cuda.synchronize()
    build_ms["kernel layouts (first product of each graph)"] = (time.perf_counter() - t0) * 1e3
    return ops, build_ms, (nd, ns, E, knn_k)


def run_step(torch, ops, comm_stream, record, exchange=None):
    """All 8 SpMMs; for N > 1 each row block is exchanged on a side stream while the next SpMM runs
    (the drug side, the disease side and the FGCN channel are independent)."""
    cur = torch.cuda.current_stream()
    for op in ops:
        op.launch(record)
        if op.y_full is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                op.shard.gather_rows(op.y_local, out=op.y_full, exchange=exchange)
    if comm_stream is not None:
        cur.wait_stream(comm_stream)


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box gives a 1-GPU job 16 of the host's 256 hardware threads)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(torch, ops, budget_s=12.0):
    """The CPU oracle (OpenMP restatement of DGL's row-parallel copy_u->sum) on the first GCMC op."""
    from oracle import oracle as O

    O.build()
    op = ops[0]
    g = op.shard.local
    indptr, indices = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
    X, ss, ds = op.X.cpu().numpy(), op.ss.cpu().numpy(), op.ds[op.shard.lo:op.shard.hi].cpu().numpy()
    threads = min(host_cores(), O.max_threads())
    O.spmm_csr(indptr[:1025], indices, None, X, ss, ds[:1024], threads=threads)  # touch / warm
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        O.spmm_csr(indptr, indices, None, X, ss, ds, threads=threads, validate=False)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    out = {"value": op.nnz / med, "unit": "edges/s", "cores": threads, "kind": "port",
           "sample": "%s (%d edges, F=%d), %d reps of the OpenMP oracle, median %.1f ms; the reference's "
                     "CPU DGL kernel itself is not installable here" % (op.name, op.nnz, F, len(times), med * 1e3)}
    # Beside it, the one call of the path the reference makes that IS runnable here: th.spmm on a sparse
    # COO tensor (layers.py:312), on the disease kNN-64 graph, on the same host cores.
    try:
        kn = next(o for o in ops if o.name == "fgcn_fwd disease-knn")
        S_ = kn.shard.local._S
        adj = torch.sparse_coo_tensor(torch.stack([S_.dst.long().cpu(), S_.src.long().cpu()]),
                                      kn.shard.local._coo_vals.cpu(), (S_.n_dst, S_.n_src))
        Xc = kn.X.cpu()
        torch.set_num_threads(threads)
        torch.spmm(adj, Xc)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            torch.spmm(adj, Xc)
            ts.append(time.perf_counter() - t0)
        out["th_spmm_coo"] = {"value": kn.nnz / sorted(ts)[1], "unit": "edges/s", "torch_threads": threads,
                              "sample": "torch.spmm(sparse_coo, X) as layers.py:312 calls it, %s, %d nnz, median of 3 = %.0f ms"
                                        % (kn.name, kn.nnz, sorted(ts)[1] * 1e3)}
        # and the strongest stock CPU path SURVEY 8(d) names: torch.sparse_csr @ X (MKL), same graph, same cores
        crow = torch.zeros(S_.n_dst + 1, dtype=torch.int64)
        crow[1:] = torch.cumsum(torch.bincount(S_.dst.long().cpu(), minlength=S_.n_dst), 0)
        order = torch.sort(S_.dst.long().cpu(), stable=True).indices
        csr = torch.sparse_csr_tensor(crow, S_.src.long().cpu()[order], kn.shard.local._coo_vals.cpu()[order], (S_.n_dst, S_.n_src))
        csr @ Xc
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            csr @ Xc
            ts.append(time.perf_counter() - t0)
        out["torch_sparse_csr"] = {"value": kn.nnz / sorted(ts)[2], "unit": "edges/s", "torch_threads": threads,
                                   "sample": "torch.sparse_csr_tensor @ X (MKL), %s, %d nnz, median of 5 = %.0f ms"
                                             % (kn.name, kn.nnz, sorted(ts)[2] * 1e3)}
    except Exception as exc:  # noqa: BLE001  (context number only)
        out.setdefault("th_spmm_coo", {"error": repr(exc)})
        out.setdefault("torch_sparse_csr", {"error": repr(exc)})
    return out


def n1_reference():
    """N = 1 throughput to quote speed-ups against inside an N > 1 run: the committed bench line of this round
    (profiles/r04_bench.json), else an earlier round's."""
    for name in ("r04_bench.json", "r03_bench.json", "r02_bench.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                line = json.load(f)
            return {"value": float(line["value"]), "ms_per_step": float(line["ms_per_step"]), "source": "profiles/" + name}
        except (OSError, ValueError, KeyError):
            continue
    return None


def committed_traffic():
    """Fabric bytes per launch of the dominant kernel pair from the committed rocprofv3 PMC passes, with
    `traffic_stale` = the kernel sources have changed since the profile was taken (sha256 of the files the
    profile recorded against the files this run launches from; no git needed on the GPU box)."""
    import hashlib

    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if "fabric_bytes_per_launch" not in t and "hbm_bytes_per_launch" in t:  # round-2 file
        t["fabric_bytes_per_launch"] = t["hbm_bytes_per_launch"]
    recorded = t.get("kernel_source_sha256_16")
    stale = True  # a profile that did not record what it was taken with cannot be vouched for
    if recorded:
        stale = False
        for name, digest in recorded.items():
            try:
                with open(os.path.join(ROOT, "dream_gnn_amd", "csrc", name), "rb") as fh:
                    stale |= hashlib.sha256(fh.read()).hexdigest()[:16] != digest
            except OSError:
                stale = True
    t["traffic_stale"] = stale
    return t


def probe_rates(torch, dev):
    """Row-gather rates of this GPU, measured now, in the SpMM kernels' access shape
    (`dgmi_probe_row_gather_f32`: 32 lanes x 16 B per 512-B row, 8 gathers in flight per lane, sums in
    registers, hash-generated row ids) — the roofs the gather kernels actually sit under."""
    from dream_gnn_amd import _lib

    groups, per_group = 1280 * 8, 2048
    out = torch.empty(groups, F, device=dev)
    rates = {}
    for key, rows, window, per_xcd in (("l2_resident_2MB_table", 4096, 4096, 0),
                                       ("l2_xcd_local_8x3.2MB", 8 * 6250, 6250, 1),
                                       ("l2_xcd_local_8x6.4MB", 8 * 12500, 12500, 1),
                                       ("infinity_cache_51MB_uniform", 100_000, 100_000, 0),
                                       ("hbm_410MB_uniform", 800_000, 800_000, 0)):
        T = torch.randn(rows, F, device=dev)
        fn = lambda: _lib.check(_lib.lib.dgmi_probe_row_gather_f32(T.data_ptr(), rows, F, groups, per_group, window, per_xcd,
                                                                  out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                                "dgmi_probe_row_gather_f32")
        ms = timeit(torch, fn, reps=5, warm=2)
        rates[key] = groups * per_group * F * 4 / ms / 1e6  # GB/s
        del T
    a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)  # 256 MB each way
    rates["hbm_copy_256MB"] = 2 * a.numel() * 4 / timeit(torch, lambda: b.copy_(a), reps=10, warm=2) / 1e6
    return {k: round(v, 1) for k, v in rates.items()}


def variants(torch, dev, ops):
    """Products beyond the 8 of the step (N = 1): each a few launches, timed by HIP events."""
    from dream_gnn_amd import graph as G, ops as O, synth

    out = {}
    # (1) the products the TRAINING step runs: edge dropout every iteration (train.py:267), kept count
    # max(1, int(E * 0.9)) per edge list.  A dropped view of a graph in the XCD-local form compacts each layout once
    # (dgmi_compact_layout_i32) and then runs the plain kernels; `on_the_fly_ms` is the round-2/3 form (keep(eid[p]) per
    # edge, product and pass) for comparison.  A training step uses every layout 3 x (L = 3 layers): `ms_amortised_L3`.
    for op in ops[:2] + ops[4:5]:
        g = op.shard.local
        keep = max(1, int(g.nnz * 0.9))
        desc = O.random_subset_select(g.nnz, keep, 12345, dev)
        kept = int(O.keep_mask(desc, g.nnz).sum().item())
        t_sel = timeit(torch, lambda: O.random_subset_select(g.nnz, keep, 12345, dev), reps=10, warm=2)
        ds = None if op.ds is None else op.ds[op.shard.lo:op.shard.hi].contiguous()
        view = g.dropped(desc)
        t_drop = timeit(torch, lambda: view.spmm(op.X, op.ss, ds, out=op.y_local))
        t_full = timeit(torch, lambda: g.spmm(op.X, op.ss, ds, out=op.y_local))
        t_compact = timeit(torch, lambda: g.dropped(desc).spmm(op.X, op.ss, ds, out=op.y_local), reps=10, warm=2) - t_drop
        old = O.COMPACT_DROPPED
        O.COMPACT_DROPPED = False
        try:
            fly = g.dropped(desc)
            t_fly = timeit(torch, lambda: fly.spmm(op.X, op.ss, ds, out=op.y_local))
        finally:
            O.COMPACT_DROPPED = old
        out["edge_dropped " + op.name] = {"kept_edges": kept, "expected_kept": keep, "ms": round(t_drop, 4),
                                          "undropped_ms": round(t_full, 4), "layout_compaction_ms": round(t_compact, 4),
                                          "ms_amortised_L3": round(t_drop + t_compact / 3, 4),
                                          "ratio_to_undropped_amortised_L3": round((t_drop + t_compact / 3) / t_full, 3),
                                          "on_the_fly_ms": round(t_fly, 4), "subset_selection_ms": round(t_sel, 4),
                                          "gedges_per_s_kept": round(kept / t_drop / 1e6, 2)}
        del view, fly
    # (2) SURVEY §8(d): the same 10 M edges with Zipf(1.2) destination degrees (longest row ~2 M edges)
    gen = torch.Generator(device=dev).manual_seed(1)
    p = 1.0 / torch.arange(1, BASE_DIS + 1, device=dev, dtype=torch.float64) ** 1.2
    dst = torch.multinomial(p / p.sum(), BASE_EDGES, replacement=True, generator=gen).to(torch.int32)
    src = torch.randint(0, BASE_DRUG, (BASE_EDGES,), generator=gen, device=dev, dtype=torch.int32)
    gz = O.CSRGraph(dst, src, BASE_DIS, BASE_DRUG)
    y = torch.empty(BASE_DIS, F, device=dev)
    cj, ci = synth.degree_norm(src, BASE_DRUG), synth.degree_norm(dst, BASE_DIS)
    t = timeit(torch, lambda: gz.spmm(ops[0].X, cj, ci, out=y))
    alg = algorithmic_bytes(BASE_EDGES, BASE_DIS, False, BASE_DRUG, BASE_DIS)
    out["zipf1.2 drug->disease"] = {"ms": round(t, 4), "max_degree": int((gz.indptr[1:] - gz.indptr[:-1]).max()),
                                    "gedges_per_s": round(BASE_EDGES / t / 1e6, 2), "alg_GBps": round(alg / t / 1e6, 1)}
    del gz, dst, src, y
    # (3) configs 2 / 3: dataset-shaped slices and kNN-4 graphs — launch-latency bound, us per call
    for cfg, blocks, widths in (("cfg2 lrssl-shape 763x681", [synth.DATASET_SHAPES["lrssl"]], (344, 128)),
                                ("cfg3 C+G merged 1256x722", [synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]], (344, 256))):
        drug, dis, labels, nd, ns = synth.dataset_shaped_pairs(blocks)
        enc = G.build_enc_graph(drug, dis, labels, nd, ns, device=dev).int()
        entry = {"train_pairs": int(drug.numel()), "calls_per_training_step": "GCMC 12 fwd + 12 bwd per-slice (6 + 6 relation-fused), FGCN 8 + 8"}
        for width in widths:
            for can in enc.canonical_etypes:
                rel = enc[can]
                x = torch.randn(rel.n_src, width, device=dev)
                cjr, cir = rel.srcdata["cj"].reshape(-1), rel.dstdata["ci"].reshape(-1)
                g = rel.csr
                o = torch.empty(rel.n_dst, width, device=dev)
                entry["slice %s (%d edges) F=%d us" % (can[1], g.nnz, width)] = round(timeit(torch, lambda: g.spmm(x, cjr, cir, out=o), reps=50) * 1e3, 2)
            fr = enc.fused_relations("disease")[0]
            x = torch.randn(fr.n_src, width, device=dev)
            o = torch.empty(fr.n_dst, width, device=dev)
            entry["relation-fused ->disease (%d edges) F=%d us" % (fr.nnz, width)] = round(timeit(torch, lambda: fr.spmm(x, None, None, out=o), reps=50) * 1e3, 2)
            # the same aggregate in COMPLEMENT form (f3): column sum - complement cells over ~8x fewer edges
            ss_, ds_ = torch.rand(fr.n_src, device=dev), torch.rand(fr.n_dst, device=dev)
            entry["relation-fused ->disease F=%d, both scales: CSR kernel us" % width] = round(timeit(torch, lambda: fr.spmm(x, ss_, ds_, out=o), reps=50) * 1e3, 2)
            comp = enc.fused_relations_complement("disease")
            if comp is not None:
                ccsr, _, i0, blockmat = comp
                B = blockmat.shape[0]
                xe = torch.cat([x, torch.zeros(B, width, device=dev)])
                coef = (blockmat * ss_.view(-1, len(enc.fused_relations("disease")[1]))[:, i0].reshape(1, -1)).contiguous()
                sse = torch.cat([ss_, torch.ones(B, device=dev)])
                R = fr.n_src // blockmat.shape[1]

                def complement_product():
                    O.colsum_rows_(xe, coef, blockmat.shape[1], R, i0)
                    ccsr.spmm(xe, sse, ds_, out=o)

                entry["relation-fused ->disease F=%d, both scales: complement form us (%d edges + column sum)" % (width, ccsr.nnz)] = \
                    round(timeit(torch, complement_product, reps=50) * 1e3, 2)
        for n in (nd, ns):
            r, c, v = synth.knn_sim_graph(n, 4, 7, dev)
            gk = O.CSRGraph(r, c, n, n, vals=v)
            for width in (768, widths[1]):
                x = torch.randn(n, width, device=dev)
                o = torch.empty(n, width, device=dev)
                entry["kNN-4 n=%d (%d nnz) F=%d us" % (n, gk.nnz, width)] = round(timeit(torch, lambda: gk.spmm(x, out=o), reps=50) * 1e3, 2)
        out[cfg] = entry
    return out


def graph_construction(torch, dev, ops):
    """Context beside the judged step: the one-time construction of config 4's kNN similarity graphs (row f4, reference
    data_loader.py:312-344: cosine top-k of 768-d rows) — the neighbour search alone, ms per call at N = 100 000."""
    from dream_gnn_amd import ops as O

    out = {"what": "dgmi_knn_cosine_topk_f32 on random unit rows, N = 100000, D = 768; ms per search (median of 3)"}
    x = torch.randn(100_000, 768, device=dev)
    xn = x / x.norm(dim=1, keepdim=True)
    for k in (4, 64):
        O.knn_cosine_topk(xn, k)
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            O.knn_cosine_topk(xn, k)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        out["knn_search_ms_k%d" % k] = round(sorted(ts)[1], 3)
    out["full_rectangle_TFLOPs"] = round(2.0 * 100_000 * 100_000 * 768 / 1e12, 2)
    return out


def edge_dropped_step(torch, dev, ops, layers=3):
    """The second headline: the step of `run_step` with ALL 8 products edge-dropped, as every training iteration of the
    reference runs them (train.py:267: augmentation is unconditional; augmentation.py:48-52,114-118: keep
    max(1, int(0.9 E)) edges per edge list, the forward relation and its reverse drawn independently).  One step =
    ONE batched subset selection over the 4 edge lists (both directions of the bipartite list + 2 kNN graphs), the
    compaction of the 8 layouts the products read, the 8 products.  Timed as a whole by HIP events; `edges` counts the
    KEPT edges the products sum.  `layers` products per layout = what an L-layer model runs per iteration."""
    from dream_gnn_amd import ops as O

    # which edge list each product's layout belongs to (build_ops: fwd A, fwd B, bwd of A, bwd of B, then fwd / bwd of each kNN graph)
    lists = [0, 1, 0, 1, 2, 2, 3, 3]
    Es = [ops[0].nnz, ops[1].nnz, ops[4].nnz, ops[6].nnz]
    keeps = [max(1, int(e * 0.9)) for e in Es]
    ds = [None if op.ds is None else op.ds[op.shard.lo:op.shard.hi].contiguous() for op in ops]
    state = {"seed": 1}

    def step(n_layers):
        state["seed"] += 1
        descs = O.random_subset_select_batch(Es, keeps, [state["seed"] * 4 + i for i in range(4)], dev)
        views = [op.shard.local.dropped(descs[lists[i]:lists[i] + 1]) for i, op in enumerate(ops)]
        for _ in range(n_layers):
            for op, v, d in zip(ops, views, ds):
                v.spmm(op.X, op.ss, d, out=op.y_local)

    kept = float(sum(keeps[lists[i]] for i in range(len(ops))))
    t1 = timeit(torch, lambda: step(1), reps=20, warm=3)
    tL = timeit(torch, lambda: step(layers), reps=10, warm=2)
    return {"what": "run_step with all 8 products edge-dropped (10 % per edge list, a new subset every step): 1 batched "
                    "selection + 8 layout compactions + 8 products",
            "kept_edges_per_step": int(kept), "ms_per_step": round(t1, 4), "value": kept / (t1 * 1e-3), "unit": "kept edges/s",
            "ms_per_step_with_%d_products_per_layout" % layers: round(tL, 4),
            "value_L%d" % layers: layers * kept / (tL * 1e-3),
            "note": "a training iteration of an L = %d model reuses each compacted layout %d x; `value_L%d` is the rate of that "
                    "schedule, `value` pays selection + compaction for ONE product per layout" % (layers, layers, layers)}


def model_steps(torch, dev):
    """Context, not the judged metric: the whole training iteration of train.py:249-300 (per-step edge dropout +
    feature noise, 3 GCMC layers + FGCN + attention + decoder, loss, backward, clip, Adam) on the dataset-shaped
    configs 2 / 3 — eager (`harness.train_step`) and recorded once as a HIP graph (`harness.CapturedTrainStep`: the
    subsets' seeds are drawn on the device, so nothing in the step needs the host).  ms per step by the wall clock."""
    import types

    from dream_gnn_amd import graph as G, harness as H, model as M, synth

    out = {}
    for cfg, blocks, width in (("cfg2 lrssl-shape", [synth.DATASET_SHAPES["lrssl"]], 128),
                               ("cfg3 C+G merged", [synth.DATASET_SHAPES["Cdataset"], synth.DATASET_SHAPES["Gdataset"]], 256)):
        try:
            drug, dis, labels, nd, ns = synth.dataset_shaped_pairs(blocks)
            batch = {"enc_graph": G.build_enc_graph(drug, dis, labels, nd, ns, device=dev).int(),
                     "dec_graph": G.build_dec_graph(drug, dis, nd, ns, device=dev).int()}
            for key, n, seed in (("drug", nd, 1), ("disease", ns, 2)):
                batch[key + "_sim_feat"] = torch.rand(n, n, device=dev)
                batch[key + "_feat"] = torch.nn.functional.normalize(torch.randn(n, 768, device=dev))
                for gname, s2 in ((key + "_graph", 0), (key + "_feature_graph", 10)):
                    r, c, v = synth.knn_sim_graph(n, 4, seed + s2, dev)
                    batch[gname] = torch.sparse_coo_tensor(torch.stack([r.long(), c.long()]), v, (n, n))
            args = types.SimpleNamespace(rating_vals=[0, 1], src_in_units=768, dst_in_units=768, gcn_agg_units=1024,
                                         gcn_out_units=width, dropout=0.3, gcn_agg_accum="sum", model_activation="leaky",
                                         share_param=True, device=None, layers=3, fdim_drug=nd, fdim_disease=ns,
                                         nhid1=768, nhid2=width, attention_dropout=0.5)
            y = labels.to(dev).float()

            def wall(fn, n=20, warm=3):
                for _ in range(warm):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3

            from dream_gnn_amd import layers as L

            torch.manual_seed(0)
            net = M.Net(args).to(dev)
            opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
            eager = wall(lambda: H.train_step(net, opt, batch, y))
            entry = {"train_pairs": int(drug.numel()), "nodes": "%dx%d" % (nd, ns), "width": width,
                     "training_step_eager_ms": round(eager, 3), "complement_form_default": L.GCMCLayer.complement_form}
            # the relation-fused aggregate in its two forms (f3: plain CSR / column sum - complement), device time of the
            # recorded training iteration (augmentation on) and of the recorded eval forward: what decides the default
            default_form = L.GCMCLayer.complement_form
            try:
                for form, flag in (("plain", False), ("complement", True)):
                    L.GCMCLayer.complement_form = flag
                    torch.manual_seed(0)
                    net = M.Net(args).to(dev)
                    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5, capturable=True)
                    step = H.CapturedTrainStep(net, opt, batch, y)
                    entry["training_step_hip_graph_replay_ms " + form] = round(wall(step, n=40), 3)
                    del step
                    net.eval()
                    fwd = lambda: net(batch["enc_graph"], batch["dec_graph"], batch["drug_graph"], batch["drug_sim_feat"],
                                      batch["drug_feat"], batch["disease_graph"], batch["disease_sim_feat"], batch["disease_feat"],
                                      batch.get("drug_feature_graph"), batch.get("disease_feature_graph"))[0]
                    with torch.no_grad():
                        side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            for _ in range(3):
                                fwd()
                        torch.cuda.current_stream().wait_stream(side)
                        torch.cuda.synchronize()
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr):
                            fwd()
                    entry["eval_forward_hip_graph_replay_ms " + form] = round(wall(gr.replay, n=40), 3)
                    del gr, net, opt
                    torch.cuda.empty_cache()
            finally:
                L.GCMCLayer.complement_form = default_form
            # the defaults: training -> plain unless forced on; eval -> complement unless forced off (layers.GCMCLayer.complement_form)
            entry["training_step_hip_graph_replay_ms"] = entry["training_step_hip_graph_replay_ms " +
                                                               ("complement" if default_form is True else "plain")]
            entry["eval_forward_hip_graph_replay_ms"] = entry["eval_forward_hip_graph_replay_ms " +
                                                              ("plain" if default_form is False else "complement")]
            out[cfg] = entry
            del batch
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 - context numbers must never cost the bench its line
            out[cfg] = {"error": repr(exc)}
    return out


def model_steps_in_child():
    """`model_steps` in a child process of its own: it records HIP graphs and trains two models — context numbers
    that must not be able to cost this process its JSON line, whatever happens in there."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--model-steps-child"], capture_output=True, text=True,
                           timeout=480, env=dict(os.environ, DGMI_SKIP_BUILD="1"))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "no result from the child (rc %d): %s" % (r.returncode, r.stderr.strip()[-300:])}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}

def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this file (one per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what
    `torch.distributed.run` would set) and return the exit code to leave with.  Called before this process has imported
    torch or made any HIP call — the children are ordinary spawns, nothing is exec'ed in place.  Rank 0's stdout is
    relayed line by line and its JSON line is printed again as the LAST line of stdout; the other ranks' stdout goes to
    stderr.  The first rank that exits non-zero ends the others (SIGTERM, SIGKILL 10 s later), and so does the time
    limit (`DGMI_BENCH_LAUNCH_TIMEOUT` seconds, default 1500): nobody waits forever on a peer that died."""
    import signal
    import subprocess
    import threading

    timeout_s = float(os.environ.get("DGMI_BENCH_LAUNCH_TIMEOUT", "1500")) if timeout_s is None else timeout_s
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(_free_port()),
               WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), DGMI_SKIP_BUILD="1",  # built by the parent, once
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs, relays, last_json = [], [], [None]

    def relay(pipe, sink, remember):
        for line in pipe:
            if remember and line.lstrip().startswith("{"):
                last_json[0] = line.rstrip("\n")
                continue  # printed once, last
            sink.write(line)
            sink.flush()

    def stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    for r in range(n):
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                             stdout=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=relay, args=(p.stdout, sys.stdout if r == 0 else sys.stderr, r == 0), daemon=True)
        t.start()
        relays.append(t)
    old = {s: signal.signal(s, lambda signum, _f: (stop_all(signum), sys.exit(128 + signum))) for s in (signal.SIGTERM, signal.SIGINT)}
    rc, deadline = 0, time.monotonic() + timeout_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1] if bad[0][1] > 0 else 1
                sys.stderr.write("bench.py self-launch: rank %d exited with %d; stopping the other ranks\n" % bad[0])
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc = 124
                sys.stderr.write("bench.py self-launch: %d ranks still running after %.0f s; stopping them\n"
                                 % (sum(c is None for c in codes), timeout_s))
                break
            time.sleep(0.1)
        if rc != 0:
            stop_all(signal.SIGTERM)
            t_kill = time.monotonic() + 10
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(0.1)
            stop_all(signal.SIGKILL)
        for p in procs:
            p.wait()
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    for t in relays:
        t.join(timeout=5)
    if last_json[0] is not None:
        print(last_json[0], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py self-launch: rank 0 printed no JSON line\n")
        rc = 1
    return rc

def launcher_rehearsal(args, dist, rank, world):
    """DGMI_BENCH_REHEARSAL=launcher: what the ranks of a self-launched job do when only the LAUNCH is under test
    (tests/test_bench_launch.py, no GPU): rendezvous over gloo at the address the launcher handed out, the same
    barrier + max-over-ranks reduction `measure` ends with, rank 0 prints a JSON line.  No product code runs: the
    product has no CPU path.  DGMI_BENCH_REHEARSAL_FAIL_RANK / _HANG_RANK make one rank exit non-zero / never finish."""
    import torch

    if os.environ.get("DGMI_BENCH_REHEARSAL_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    dist.init_process_group("gloo")
    if os.environ.get("DGMI_BENCH_REHEARSAL_HANG_RANK") == str(rank):
        time.sleep(3600)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = torch.tensor([1.0])
    dist.all_reduce(ranks)
    if rank == 0:
        print("rank 0 of %d: rendezvous at %s:%s" % (world, os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"]), flush=True)
        print(json.dumps({"rehearsal": "launcher", "n_gpus": world, "ranks_seen": int(ranks.item()), "max_over_ranks": float(t.item()),
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "--model-steps-child" in sys.argv:
        import torch

        import dream_gnn_amd  # noqa: F401
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        print(json.dumps(model_steps(torch, dev)), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # >= 50 launches per product after >= 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra products and probes (profiling runs)")
    ap.add_argument("--scale", choices=["nodes", "edges"], default="nodes",
                    help="what grows with N in the primary run: nodes and edges (SURVEY §8d's config 5, default) "
                         "or edges over config 4's node set; the other one is measured too")
    args = ap.parse_args()

    # Native pieces first: make decides freshness, children are spawned before this process
    # touches the GPU (and before a profiler-preloaded runtime is initialised any further).
    import __graft_entry__

    __graft_entry__.ensure_built()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL / device-tensor sharing needs on this stack
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be one.  Nothing in this process has touched the GPU (torch is not even imported yet).
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if os.environ.get("DGMI_BENCH_REHEARSAL") == "launcher":
        return launcher_rehearsal(args, dist, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    # One rank per GPU.  (Rehearsal of the N>1 path on a 1-GPU box: DGMI_DIST_BACKEND=gloo lets
    # several ranks share cuda:0 and stages the exchange through the host.)
    backend = os.environ.get("DGMI_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, n_dev))
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import dream_gnn_amd  # noqa: F401  (fails loudly if libdgmi.so is missing)

    comm_stream = torch.cuda.Stream() if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(ops, steps, warmup, exchange=None):
        for op in ops:
            op.events = []
        for _ in range(warmup):
            run_step(torch, ops, comm_stream, record=False, exchange=exchange)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step(torch, ops, comm_stream, record=True, exchange=exchange)
        barrier()
        elapsed = time.perf_counter() - t0
        edges = torch.tensor([float(sum(op.nnz for op in ops))], dtype=torch.float64, device=dev)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            dist.all_reduce(edges)
        return elapsed, float(edges.item())

    def pick_exchange(ops):
        """The row-block exchange form (dream_gnn_amd.shard.choose_exchange).  Default: the RCCL all-gather — the
        well-trodden collective, and the only form the judged reading depends on.  Opt-in: DGMI_EXCHANGE=auto times
        the all-gather, then the all-links batched point-to-point form inside try/except (3 steps each) and keeps
        the faster (a form that raises on any rank is dropped on every rank; a HANG inside an untried collective
        cannot be guarded, which is why this is not the default); DGMI_EXCHANGE=direct forces the point-to-point form."""
        from dream_gnn_amd import shard as S

        mode = os.environ.get("DGMI_EXCHANGE", "allgather")
        return S.choose_exchange(lambda ex: measure(ops, 3, 1, exchange=ex)[0] / 3, dev, world, mode=mode)

    def reading(ops, elapsed, edges, steps, exchange):
        """What a weak-scaling reading reports beside its throughput: per-rank compute, the exchange time the
        overlap did not hide, bytes received per rank, and the step time / speed-up DESIGN §6's model predicts
        from the measured compute (checkable against the measured ones in the same object)."""
        from dream_gnn_amd import shard as S

        comp = [sum(a.elapsed_time(b) for a, b in op.events) / max(len(op.events), 1) * 1e-3 for op in ops]  # s per product
        recv = [float((op.shard.n_dst - (op.shard.hi - op.shard.lo)) * F * 4) for op in ops]
        t = torch.tensor(comp + [sum(comp)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comp_max, comp_sum = [float(v) for v in t[:-1]], float(t[-1])
        ms_step = elapsed / steps * 1e3
        n1 = n1_reference()
        out = {"ms_per_step": ms_step, "per_rank_compute_ms_per_step": round(comp_sum * 1e3, 4),
               "exchange_ms_not_hidden": round(ms_step - comp_sum * 1e3, 4),
               "per_rank_recv_MB_per_step": round(sum(recv) / 1e6, 1), "exchange": exchange,
               "model": "DESIGN 6: compute stream runs the products back to back; each exchange is queued on one side "
                        "stream when its product ends; allgather = recv bytes over ONE %g GB/s xGMI link (ring), direct = "
                        "over min(N-1, %d) links" % (S.XGMI_LINK_GBS, S.XGMI_LINKS),
               "predicted_ms_per_step": {f: round(S.predict_step_seconds(comp_max, recv, world, f) * 1e3, 4)
                                         for f in ("allgather", "direct")}}
        if n1 is not None:
            out["n1_reference"] = n1
            out["speedup_vs_n1_reference"] = round(edges * steps / elapsed / n1["value"], 3)
            out["predicted_speedup"] = {f: round(edges / (v * 1e-3) / n1["value"], 3) for f, v in out["predicted_ms_per_step"].items()}
        return out

    progress("building the workload (%d rank%s)" % (world, "" if world == 1 else "s, %s-scaled" % args.scale))
    ops, build_ms, (nd, ns, E, knn_k) = build_ops(torch, rank, world, dev, args.scale if world > 1 else "edges")
    progress("workload built: bipartite %dx%d, %d edges + kNN-%d; timing %d + %d steps" % (nd, ns, E, knn_k, args.warmup, args.steps))
    exchange, exchange_timing = pick_exchange(ops)
    elapsed, edges_per_step = measure(ops, args.steps, args.warmup, exchange=exchange)
    progress("timed region done: %.3f ms per step" % (elapsed / args.steps * 1e3))
    primary_reading = reading(ops, elapsed, edges_per_step, args.steps, exchange) if world > 1 else None

    # per-kernel HIP-event time on the launch stream (rank 0's launches)
    probes = probe_rates(torch, dev) if (world == 1 and not args.no_variants) else None

    def l2_roof(op):
        """The probe rate of the regime this product's gather runs in (GB/s), or None."""
        if probes is None:
            return None, None
        if op.table_bytes <= 6 << 20:
            return "l2_resident_2MB_table", probes["l2_resident_2MB_table"]
        if op.table_bytes <= 30 << 20:
            return "l2_xcd_local_8x3.2MB", probes["l2_xcd_local_8x3.2MB"]
        if op.table_bytes <= 64 << 20:
            return "l2_xcd_local_8x6.4MB", probes["l2_xcd_local_8x6.4MB"]
        return "hbm_410MB_uniform", probes["hbm_410MB_uniform"]

    per_op = {}
    dom_t = dom_b = dom_c = 0.0
    dom_n = 0
    dom_roofs = []
    for op in ops:
        ms = [a.elapsed_time(b) for a, b in op.events]
        avg = sum(ms) / len(ms)
        srt = sorted(ms)
        roof_name, roof = l2_roof(op)
        entry = {"avg_ms": round(avg, 4), "median_ms": round(srt[len(srt) // 2], 4),
                 "p10_ms": round(srt[len(srt) // 10], 4), "p90_ms": round(srt[(len(srt) * 9) // 10], 4),
                 "gedges_per_s": round(op.nnz / avg / 1e6, 2), "alg_bytes": op.bytes, "compulsory_bytes": op.compulsory,
                 "alg_GBps": round(op.bytes / avg / 1e6, 1),
                 "frac_hbm_contract": round(op.bytes / avg / 1e6 / HBM_PEAK_GBS, 3)}
        if roof is not None:
            entry["frac_of_l2_gather_probe"] = round(op.bytes / avg / 1e6 / roof, 3)
            entry["l2_gather_probe"] = roof_name
        per_op[op.name] = entry
        if op.dominant:
            dom_t += sum(ms) * 1e-3
            dom_b += op.bytes * len(ms)
            dom_c += op.compulsory * len(ms)
            dom_n += len(ms)
            if roof is not None:
                dom_roofs.append(roof)

    other = None
    if world > 1:  # the other weak-scaling reading, same run, equal standing
        alt = "nodes" if args.scale == "edges" else "edges"
        del ops
        torch.cuda.empty_cache()
        ops2 = None
        progress("building the %s-scaled workload" % alt)
        try:  # a rank that cannot build its shards must not leave the others inside a collective
            ops2, _, (nd2, ns2, E2, k2) = build_ops(torch, rank, world, dev, alt)
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("rank %d: %s-scaled workload not built: %r\n" % (rank, alt, exc))
        ok = torch.tensor([1.0 if ops2 is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            el2, edges2 = measure(ops2, args.steps, args.warmup, exchange=exchange)
            progress("%s-scaled reading done: %.3f ms per step" % (alt, el2 / args.steps * 1e3))
            other = {"workload": "bipartite %dx%d, %d edges + kNN-%d" % (nd2, ns2, E2, k2),
                     "value": edges2 * args.steps / el2, "unit": "edges/s",
                     "steps": args.steps, "edges_per_step": int(edges2)}
            other.update(reading(ops2, el2, edges2, args.steps, exchange))
        ops = ops2 or []

    if rank == 0:
        achieved = dom_b / dom_t / 1e9
        traffic = committed_traffic() if world == 1 else None  # the PMC passes profiled the N = 1 products
        avg_launch_s = dom_t / dom_n
        roofline = {
            "bound": "hbm", "kernel": DOMINANT, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_definition": "contract: SURVEY §8(d) ALGORITHMIC bytes / launch time / HBM peak. The per-edge row re-reads "
                               "it counts are served by L2 / Infinity Cache, so it is not bounded by 1; the bounded readings "
                               "are hbm_traffic_frac and l2_gather.frac below",
            "traffic": None if traffic is None else traffic.get("fabric_bytes_per_launch"),
            "traffic_is": "bytes crossing the L2 -> memory-side (fabric) boundary per launch pair, Infinity-Cache hits included; "
                          "the HBM-only share is not observable on gfx950 (no MALL / UMC counter): compulsory_bytes <= HBM bytes <= traffic",
            "traffic_source": None if traffic is None else
            "committed rocprofv3 PMC passes, not this run: %s; %s" % (traffic.get("source"), traffic.get("correction")),
            "traffic_profiled_at_commit": None if traffic is None else traffic.get("profiled_at_commit"),
            "traffic_stale": None if traffic is None else traffic.get("traffic_stale"),
            "dram_targeted_share": None if traffic is None else traffic.get("dram_targeted_share"),
            "launches": dom_n, "avg_launch_ms": avg_launch_s * 1e3,
            "alg_bytes_per_launch": dom_b / dom_n, "compulsory_bytes_per_launch": dom_c / dom_n,
            "frac_vs_measured_copy_6.29TBps_contract_bytes": achieved / HBM_COPY_GBS,
        }
        if traffic is not None and traffic.get("fabric_bytes_per_launch"):
            roofline["hbm_traffic_frac"] = traffic["fabric_bytes_per_launch"] / avg_launch_s / 1e9 / HBM_PEAK_GBS
            roofline["traffic_over_compulsory"] = traffic["fabric_bytes_per_launch"] / (dom_c / dom_n)
        if dom_roofs:
            peak = sum(dom_roofs) / len(dom_roofs)
            roofline["l2_gather"] = {"bound": "l2_gather", "peak": peak, "achieved": achieved, "unit": "GB/s",
                                     "frac": achieved / peak,
                                     "peak_source": "dgmi_probe_row_gather_f32 run in this process: 512-B rows gathered by hash "
                                                    "ids from 8 XCD-local windows of the product's slice size (mean over the "
                                                    "dominant products' regimes); the product additionally streams 40 MB of "
                                                    "indices and 0.6 GB of plane scratch"}
        out = {
            # BASELINE.json's metric, verbatim; `value` is its edges/sec half, the achieved GB/s
            # half is `roofline.achieved`
            "metric": "GCMC+FGCN SpMM edges/sec and achieved HBM GB/s at 1/2/4/8 MI355X",
            "value": edges_per_step * args.steps / elapsed,
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "BASELINE config %s: bipartite %dx%d, %d edges + kNN-%d sim graphs, F=%d; "
                            "one step = 4 GCMC (copy_u->sum, cj/ci fused) + 4 FGCN (weighted) SpMMs, fwd+bwd"
                            % ("4" if world == 1 else "5 (config 4 weak-scaled x%d in %s)" % (world, "nodes and edges" if args.scale == "nodes" else "edges"),
                               nd, ns, E, knn_k, F),
                "weak_scaling_in": None if world == 1 else args.scale,
                "edges_per_step": int(edges_per_step),
                "parallelism": "single GPU" if world == 1 else
                               "%d ranks, destination-row-aligned nnz-balanced edge partition, row-block exchange over RCCL "
                               "(%s), overlapped with the next SpMM" % (world, exchange),
                "graph_build_ms": {k: round(v, 2) for k, v in build_ms.items()},
            },
            "roofline": roofline,
            "kernels": per_op,
        }
        if world > 1:  # ms per step by exchange form: both when DGMI_EXCHANGE=auto timed them, else the judged run's own
            out["config"]["exchange_ms_per_step"] = exchange_timing if exchange_timing is not None else \
                {exchange: round(elapsed / args.steps * 1e3, 3), "note": "the timed run itself; DGMI_EXCHANGE=auto (opt-in) also times the all-links form"}
        if primary_reading is not None:
            out["node_scaled" if args.scale == "nodes" else "edge_scaled"] = primary_reading
            out["config"]["six_x_claim"] = (
                "north_star's >= 6x aggregate at 8 GPUs is carried by the EDGE-scaled reading (config 4's node set, "
                "N x the edges: per-rank tables stay 26-51 MB, L2-sliceable); the NODE-scaled reading (SURVEY 8d's "
                "config 5, this line's `value` unless --scale edges) cannot reach it: each rank gathers from a replicated "
                "205-410 MB table at the HBM gather roof (~7.4 TB/s algorithmic, 0.93 of the 8 TB/s peak), ~2x slower "
                "than the cache-resident N = 1 product, so it tops out near 4.2-4.4x before any exchange - DESIGN 6")
        if probes is not None:
            out["probes_GBps"] = probes
        if other is not None:
            out["edge_scaled" if args.scale == "nodes" else "node_scaled"] = other
        if world == 1 and not args.no_variants:
            progress("variants, edge-dropped step, model steps (context beside the judged step)")
            try:  # extra products beside the judged step: a failure here must not cost the line
                out["variants"] = variants(torch, dev, ops)
            except Exception as exc:  # noqa: BLE001
                out["variants"] = {"error": repr(exc)}
            try:
                out["edge_dropped_step"] = edge_dropped_step(torch, dev, ops)
            except Exception as exc:  # noqa: BLE001
                out["edge_dropped_step"] = {"error": repr(exc)}
            try:
                out["graph_construction"] = graph_construction(torch, dev, ops)
            except Exception as exc:  # noqa: BLE001
                out["graph_construction"] = {"error": repr(exc)}
            out["model_steps"] = model_steps_in_child()
        if world == 1 and not args.no_cpu_baseline:
            progress("cpu_baseline (bounded sample on the host cores)")
            try:
                out["cpu_baseline"] = cpu_baseline(torch, ops)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
