"""Single-GPU prediction of bench.py's N = 2 / 4 / 8 lines (no 8-GPU node is available to this pipeline).

For each world size and both weak-scaling readings, builds RANK 0's shard of every product exactly as
bench.py does (bench.build_ops), times the 8 local products on this GPU (per-rank compute: the graph is
uniform, every rank's shard is statistically the same), and applies DESIGN 6's exchange model
(dream_gnn_amd.shard.predict_step_seconds) for both exchange forms.  Writes profiles/r04_scale_prediction.json;
bench.py prints the same model's numbers beside the measured ones when it does run on N GPUs.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from dream_gnn_amd import shard as S

dev = torch.device("cuda:0")
F = bench.F
rows = []
n1_step = None
worlds = [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]
for world in worlds:
    for scale in (("edges",) if world == 1 else ("nodes", "edges")):
        ops, _, (nd, ns, E, k) = bench.build_ops(torch, 0, world, dev, scale)
        comp = [bench.timeit(torch, lambda op=op: op.launch(False), reps=20, warm=3) * 1e-3 for op in ops]
        recv = [float((op.shard.n_dst - (op.shard.hi - op.shard.lo)) * F * 4) for op in ops]
        edges_rank = sum(op.nnz for op in ops)
        entry = {"world": world, "scale": scale, "workload": "bipartite %dx%d, %d edges + kNN-%d" % (nd, ns, E, k),
                 "edges_per_rank_per_step": edges_rank,
                 "per_product_ms": {op.name: round(c * 1e3, 4) for op, c in zip(ops, comp)},
                 "per_rank_compute_ms_per_step": round(sum(comp) * 1e3, 4),
                 "per_rank_recv_MB_per_step": round(sum(recv) / 1e6, 1)}
        if world == 1:
            n1_step = sum(comp)
            n1_edges = edges_rank
        for form in ("allgather", "direct"):
            t = S.predict_step_seconds(comp, recv, world, form)
            entry["predicted_ms_per_step_" + form] = round(t * 1e3, 4)
            entry["exchange_ms_total_" + form] = round(sum(S.exchange_seconds(b, world, form) for b in recv) * 1e3, 4)
            if n1_step:
                entry["predicted_speedup_" + form] = round((edges_rank * world / t) / (n1_edges / n1_step), 3)
        rows.append(entry)
        print(json.dumps(entry), flush=True)
        del ops
        torch.cuda.empty_cache()
with open(os.path.join(ROOT, "gpurun_out", "r04_scale_prediction.json"), "w") as f:  # copied to profiles/ by hand
    json.dump({"gpu": torch.cuda.get_device_name(0), "model": S.predict_step_seconds.__doc__, "link_GBps": S.XGMI_LINK_GBS,
               "links": S.XGMI_LINKS, "rows": rows}, f, indent=1)
