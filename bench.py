#!/usr/bin/env python3
"""bench.py — GCMC+FGCN SpMM edges/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (N = 1): BASELINE config 4 — synthetic bipartite 100k drugs x 50k diseases with
10M edges plus kNN-64 similarity graphs on both node sets, F = 128, fp32.  One *step* is one
pass of the message-passing hot path over those graphs, forward and backward:

    GCMC   copy_u->sum with the cj/ci scalings fused   (reference layers.py:224-234)
             drug->disease, disease->drug, and the transpose (autograd) of each      4 launches
    FGCN   th.spmm(adj, support)                       (reference layers.py:312)
             drug-kNN, disease-kNN, and the transpose of each                         4 launches

`value` = edges processed per second over the whole job, inputs resident in HBM.  For N > 1
the problem is weak-scaled in EDGES — BASELINE config 5 names only "80M-edge bipartite": the node
set stays config 4's 100k x 50k and the bipartite graph has N x 10M distinct edges, the kNN
graphs k = 64 N neighbours, so every rank processes the same number of edges as the N = 1 run.
(In this domain edges grow with density at a fixed node set: the reference's encoder graph is
all drug x disease pairs.)  Every rank owns 1/N of the destination rows of every graph and all
their in-edges, each local SpMM is followed by the all-gather of its row block over RCCL, and
the time is the max over ranks (dream_gnn_amd/shard.py).  The other weak-scaling reading —
N x nodes AND N x edges (800k x 400k at N = 8), where the replicated feature table outgrows the
L2-sliced kernel and the all-gather payload grows N-fold — is measured in the same run and
reported under `node_scaled_variant`; `--scale nodes` makes it the primary.

The JSON line also carries `roofline` for the dominant kernel (the unweighted scaled F=128
SpMM — an XCD-local gather kernel plus its plane-reduce kernel: algorithmic bytes / HIP-event
time of the pair on the launch stream, against the 8 TB/s HBM peak) and
`cpu_baseline` (the OpenMP CPU oracle timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
F = 128
BASE_DRUG, BASE_DIS, BASE_EDGES, KNN_K = 100_000, 50_000, 10_000_000, 64
# The GCMC product at F=128 on config 4's 25-51 MB feature tables runs as the XCD-local pair:
# gather kernel (LPR=32, unweighted, src scale) + the 8-plane reduce (dst scale).  Its time is
# taken with HIP events around the pair; rocprofv3's averages of the two kernels add up to it.
DOMINANT = "spmm_sliced_vec4_kernel<32,false,true> + reduce_planes_kernel<true,8>"


def algorithmic_bytes(nnz, n_rows, weighted, n_scales_src=0, n_scales_dst=0):
    """nnz*(4F + 4 + 4w) + N_dst*4F + (N_dst+1)*4 (+ fused scale vectors)."""
    return nnz * (4 * F + 4 + (4 if weighted else 0)) + n_rows * 4 * F + (n_rows + 1) * 4 \
        + 4 * n_scales_src + 4 * n_scales_dst


class Op:
    """One SpMM of the step: this rank's row block of Y = diag(ds) A diag(ss) X (+ all-gather)."""

    def __init__(self, name, shard, X, ss, ds, weighted, dominant):
        self.name, self.shard, self.X, self.ss, self.ds = name, shard, X, ss, ds
        self.weighted, self.dominant = weighted, dominant
        rows = shard.hi - shard.lo
        self.y_local = torch.empty((rows, F), dtype=torch.float32, device=X.device)
        self.y_full = None if shard.world == 1 else torch.empty((shard.n_dst, F), dtype=torch.float32, device=X.device)
        self.nnz = shard.nnz
        self.bytes = algorithmic_bytes(self.nnz, rows, weighted,
                                       0 if ss is None else ss.numel(), 0 if ds is None else rows)
        self.events = []

    def launch(self, record):
        if record:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        self.shard.spmm_local(self.X, self.ss, self.ds, out=self.y_local)
        if record:
            b.record()
            self.events.append((a, b))


def build_ops(rank, world, dev, scale="edges"):
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

    def even(n):
        return torch.arange(world + 1, dtype=torch.int64) * (n // world)

    t0 = time.perf_counter()
    ops = []
    # GCMC: drug -> disease (rows = diseases) and disease -> drug (rows = drugs); backward = reversed edges
    fwd_ds = S.RowShard(dis, drug, ns, nd, even(ns), rank)
    fwd_sd = S.RowShard(drug, dis, nd, ns, even(nd), rank)
    ops.append(Op("gcmc_fwd drug->disease", fwd_ds, x_drug, cj_drug, ci_dis, False, True))
    ops.append(Op("gcmc_fwd disease->drug", fwd_sd, x_dis, ci_dis, cj_drug, False, True))
    # dX = diag(cj) A^T diag(ci) dY: rows = sources; the reversed edge list partitioned by source range
    ops.append(Op("gcmc_bwd drug->disease", S.RowShard(drug, dis, nd, ns, even(nd), rank), x_dis, ci_dis, cj_drug, False, True))
    ops.append(Op("gcmc_bwd disease->drug", S.RowShard(dis, drug, ns, nd, even(ns), rank), x_drug, cj_drug, ci_dis, False, True))
    del drug, dis
    # FGCN: row-normalised symmetrised kNN-64 graphs (values are not symmetric -> real transpose for bwd)
    for tag, n, x, seed in (("drug", nd, x_drug, 21), ("disease", ns, x_dis, 22)):
        r, c, v = synth.knn_sim_graph(n, knn_k, seed, dev)
        ops.append(Op("fgcn_fwd %s-knn" % tag, S.RowShard(r, c, n, n, even(n), rank, vals=v), x, None, None, True, False))
        ops.append(Op("fgcn_bwd %s-knn" % tag, S.RowShard(c, r, n, n, even(n), rank, vals=v), x, None, None, True, False))
        del r, c, v
    for op in ops:  # lazily built layouts (sliced CSR, plans) exist before anything is timed, whatever --warmup says
        op.launch(False)
    torch.cuda.synchronize()
    return ops, (time.perf_counter() - t0) * 1e3, (nd, ns, E, knn_k)


def run_step(ops, comm_stream, record, lanes=None):
    """All 8 SpMMs; for N > 1 each row block is all-gathered on a side stream while the next
    SpMM runs (the drug side, the disease side and the FGCN channel are independent).
    `lanes`: optional extra compute streams; the products are dealt round-robin over
    [current stream] + lanes (they are independent, as the two node types and the FGCN channel are
    in the model), so one product's plane-reduce overlaps the next product's gather."""
    cur = torch.cuda.current_stream()
    streams = [cur] + list(lanes or [])
    if len(streams) > 1:
        start = torch.cuda.Event()
        start.record(cur)
        for st in streams[1:]:
            st.wait_event(start)
    for i, op in enumerate(ops):
        st = streams[i % len(streams)]
        with torch.cuda.stream(st):
            op.launch(record)
            if op.y_full is not None:
                ev = torch.cuda.Event()
                ev.record(st)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ev)
                    op.shard.gather_rows(op.y_local, out=op.y_full)
    for st in streams[1:]:
        cur.wait_stream(st)
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


def cpu_baseline(ops, budget_s=12.0):
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
    except Exception as exc:  # noqa: BLE001  (context number only)
        out["th_spmm_coo"] = {"error": repr(exc)}
    return out


def committed_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale", choices=["edges", "nodes"], default="edges",
                    help="what grows with N: edges over config 4's node set (default) or nodes and edges")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (see module docstring)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    # One rank per GPU.  (Rehearsal of the N>1 path on a 1-GPU box: DGMI_DIST_BACKEND=gloo lets
    # several ranks share cuda:0 and stages the all-gather through the host.)
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

    if not os.path.exists(os.path.join(ROOT, "dream_gnn_amd", "libdgmi.so")):  # snapshot without build products
        if local_rank == 0:
            import __graft_entry__

            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    import dream_gnn_amd  # noqa: F401  (fails loudly if libdgmi.so is missing)

    comm_stream = torch.cuda.Stream() if world > 1 else None
    n_lanes = int(os.environ.get("DGMI_BENCH_STREAMS", "1"))
    lanes = [torch.cuda.Stream() for _ in range(n_lanes - 1)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(ops, steps, warmup):
        for _ in range(warmup):
            run_step(ops, comm_stream, record=False, lanes=lanes)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step(ops, comm_stream, record=True, lanes=lanes)
        barrier()
        elapsed = time.perf_counter() - t0
        edges = torch.tensor([float(sum(op.nnz for op in ops))], dtype=torch.float64, device=dev)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            dist.all_reduce(edges)
        return elapsed, float(edges.item())

    ops, build_ms, (nd, ns, E, knn_k) = build_ops(rank, world, dev, args.scale)
    elapsed, edges_per_step = measure(ops, args.steps, args.warmup)

    # per-kernel HIP-event time on the launch stream (rank 0's launches)
    per_op = {}
    dom_t = dom_b = 0.0
    dom_n = 0
    for op in ops:
        ms = [a.elapsed_time(b) for a, b in op.events]
        avg = sum(ms) / len(ms)
        srt = sorted(ms)
        per_op[op.name] = {"avg_ms": round(avg, 4), "median_ms": round(srt[len(srt) // 2], 4),
                           "p10_ms": round(srt[len(srt) // 10], 4), "p90_ms": round(srt[(len(srt) * 9) // 10], 4),
                           "gedges_per_s": round(op.nnz / avg / 1e6, 2), "alg_GBps": round(op.bytes / avg / 1e6, 1)}
        if op.dominant:
            dom_t += sum(ms) * 1e-3
            dom_b += op.bytes * len(ms)
            dom_n += len(ms)

    other = None
    if world > 1:  # the other weak-scaling reading, same run, fewer steps
        first_op = ops[0]
        del ops
        torch.cuda.empty_cache()
        alt = "nodes" if args.scale == "edges" else "edges"
        ops2 = None
        try:  # a rank that cannot build its shards must not leave the others inside a collective
            ops2, _, (nd2, ns2, E2, k2) = build_ops(rank, world, dev, alt)
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("rank %d: %s-scaled variant not built: %r\n" % (rank, alt, exc))
        ok = torch.tensor([1.0 if ops2 is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            steps2 = max(3, args.steps // 4)
            el2, edges2 = measure(ops2, steps2, min(2, args.warmup))
            other = {"scale": alt, "workload": "bipartite %dx%d, %d edges + kNN-%d" % (nd2, ns2, E2, k2),
                     "value": edges2 * steps2 / el2, "unit": "edges/s", "ms_per_step": el2 / steps2 * 1e3,
                     "steps": steps2}
        ops = [first_op]
        del ops2

    if rank == 0:
        achieved = dom_b / dom_t / 1e9
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
                            % ("4" if world == 1 else "5 (config 4 weak-scaled in %s x%d)" % (args.scale, world),
                               nd, ns, E, knn_k, F),
                "weak_scaling_in": args.scale,
                "edges_per_step": int(edges_per_step),
                "parallelism": "single GPU" if world == 1 else
                               "%d ranks, destination-row-aligned edge partition, all-gather of row blocks over RCCL, "
                               "overlapped with the next SpMM" % world,
                "csr_build_ms_all_graphs": round(build_ms, 1),
            },
            "roofline": {
                "bound": "hbm", "kernel": DOMINANT, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": committed_traffic(),
                "launches": dom_n, "avg_launch_ms": dom_t / dom_n * 1e3,
                "alg_bytes_per_launch": dom_b / dom_n,
                "note": "algorithmic gather bytes (plane scratch traffic of the XCD-local kernel not counted as "
                        "useful); per-edge row re-reads are served by the XCD's L2 / Infinity Cache, so frac "
                        "exceeds the HBM-only bound (see DESIGN.md)",
            },
            "kernels": per_op,
        }
        if other is not None:
            out["node_scaled_variant" if other["scale"] == "nodes" else "edge_scaled_variant"] = other
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ops)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
