"""RCCL on the GPU box (one MI355X: world_size 1).  The N > 1 bench path has only ever run on gloo; what one GPU can
still check is that the RCCL calls that path makes — ``init_process_group("nccl", device_id=...)``, barrier,
all-reduce (MAX / SUM), ``all_gather_into_tensor`` and one grouped ``batch_isend_irecv`` (a send to / receive from the
rank itself) — execute on this ROCm stack with the HIP kernels' tensors, and that a ``ShardedRelation`` driven through
them gives the single-process product.  The exchange ACROSS GPUs stays untested here (8-GPU runs are the driver's)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        _checks(dist, dev, q)
    except Exception as exc:  # noqa: BLE001 - reported to the parent instead of a queue timeout
        q.put({"error": repr(exc)})
    finally:
        dist.destroy_process_group()


def _checks(dist, dev, q):
    if True:
        from dream_gnn_amd import ops, shard

        out = {}
        dist.barrier()
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t)
        out["all_reduce"] = float(t.item())

        rng = np.random.default_rng(11)
        n_dst, n_src, E, F = 700, 500, 20000, 64
        dst = torch.from_numpy(rng.integers(0, n_dst, E)).to(dev)
        src = torch.from_numpy(rng.integers(0, n_src, E)).to(dev)
        val = torch.from_numpy(rng.standard_normal(E).astype(np.float32)).to(dev)
        X = torch.from_numpy(rng.standard_normal((n_src, F)).astype(np.float32)).to(dev).requires_grad_(True)
        dY = torch.from_numpy(rng.standard_normal((n_dst, F)).astype(np.float32)).to(dev)
        ss = torch.from_numpy(rng.uniform(0.2, 1, n_src).astype(np.float32)).to(dev)
        ds = torch.from_numpy(rng.uniform(0.2, 1, n_dst).astype(np.float32)).to(dev)

        rel = shard.ShardedRelation(dst, src, n_dst, n_src, vals=val)  # rank / world from the process group
        y = rel(X, ss, ds)
        y.backward(dY)
        g = ops.CSRGraph(dst.int(), src.int(), n_dst, n_src, vals=val)
        y_ref = g.spmm(X.detach(), ss, ds)
        dx_ref = g.spmm_t(dY, ss, ds)
        out["fwd_equal"] = bool(torch.equal(y.detach(), y_ref))
        out["bwd_equal"] = bool(torch.equal(X.grad, dx_ref))

        # the exchange calls themselves, on RCCL: all-gather of the one block, and one grouped send / receive
        sh = rel.fwd
        y_local = sh.spmm_local(X.detach(), ss, ds)
        full = sh.gather_rows(y_local, exchange="allgather")
        out["allgather_equal"] = bool(torch.equal(full, y_ref))
        recv = torch.empty_like(y_local)
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, y_local.contiguous(), 0), dist.P2POp(dist.irecv, recv, 0)])
        for r in reqs:
            r.wait()
        torch.cuda.synchronize()
        out["p2p_equal"] = bool(torch.equal(recv, y_local))
        out["choose"] = shard.choose_exchange(lambda form: 0.0, dev, 1)
        out["backend"] = dist.get_backend()
        q.put(out)


def test_rccl_calls_of_the_sharded_path_run_on_one_gpu():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    try:
        res = q.get(timeout=240)
        p.join(timeout=60)
    finally:
        if p.is_alive():  # a collective that never returns must not outlive the test
            p.kill()
            p.join(timeout=30)
    assert "error" not in res, res
    assert p.exitcode == 0
    assert res["backend"] == "nccl" and res["all_reduce"] == 3.5
    assert res["fwd_equal"] and res["bwd_equal"] and res["allgather_equal"] and res["p2p_equal"], res
    assert res["choose"] == (None, None)


def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher in the environment, on the GPU box: the parent spawns both ranks
    (they share cuda:0 and exchange over gloo, staged through the host — a rehearsal of the N > 1 schedule, not a
    scaling number), the real sharded workload runs, and the line that comes out last carries what an N > 1 reading
    must: both weak-scaling readings with per-rank compute, un-hidden exchange, the model's prediction."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"DGMI_DIST_BACKEND": "gloo", "DGMI_BENCH_LAUNCH_TIMEOUT": "420"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=480, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["parallelism"].startswith("2 ranks") and "(allgather)" in line["config"]["parallelism"]
    for key in ("node_scaled", "edge_scaled"):
        rd = line[key]
        for f in ("per_rank_compute_ms_per_step", "exchange_ms_not_hidden", "per_rank_recv_MB_per_step", "predicted_ms_per_step"):
            assert f in rd, (key, f)
        assert "predicted_speedup" in rd
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_self_launch_2rank_gloo.json"), "w") as f:
        f.write(json.dumps(line) + "\n")
