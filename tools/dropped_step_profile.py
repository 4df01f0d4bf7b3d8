"""The edge-dropped step of bench.py (`edge_dropped_step`: 1 batched subset selection + 8 layout compactions + 8 products
over the compacted layouts) in a loop of its own, for rocprofv3 (`--kernel-trace --stats`, or one `--pmc` pass):

    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/dropped_step_profile.py [steps] [on_the_fly]

`on_the_fly` = 1 runs the same step with DGMI_COMPACT_DROPPED off (the round-2/3 form: keep(eid[p]) per edge, product and
pass) so that the two forms' kernels can be compared counter by counter."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DGMI_SKIP_BUILD", "1")
import __graft_entry__

__graft_entry__.ensure_built()
import torch

import bench
from dream_gnn_amd import ops as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
O.COMPACT_DROPPED = not (len(sys.argv) > 2 and sys.argv[2] == "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops, _, _ = bench.build_ops(torch, 0, 1, dev, "edges")
lists = [0, 1, 0, 1, 2, 2, 3, 3]
Es = [ops[0].nnz, ops[1].nnz, ops[4].nnz, ops[6].nnz]
keeps = [max(1, int(e * 0.9)) for e in Es]
ds = [None if op.ds is None else op.ds[op.shard.lo:op.shard.hi].contiguous() for op in ops]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(steps + 3):
    if it == 3:
        torch.cuda.synchronize()
        a.record()
    descs = O.random_subset_select_batch(Es, keeps, [it * 4 + i for i in range(4)], dev)
    views = [op.shard.local.dropped(descs[lists[i]:lists[i] + 1]) for i, op in enumerate(ops)]
    for op, v, d in zip(ops, views, ds):
        v.spmm(op.X, op.ss, d, out=op.y_local)
b.record()
torch.cuda.synchronize()
print("edge-dropped step (%s): %.4f ms per step over %d steps" % ("compacted layouts" if O.COMPACT_DROPPED else "on the fly",
                                                                  a.elapsed_time(b) / steps, steps), flush=True)
