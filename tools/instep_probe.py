"""Why a product runs slower inside bench.py's step than in a loop of its own: each of the 8 products timed alone,
inside the step, and directly behind each of the other products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
ops, _, _ = bench.build_ops(torch, 0, 1, dev, "nodes")


def timed(seq, target, reps=30):
    """median ms of ops[target] when the stream runs `seq` over and over"""
    for _ in range(3):
        for i in seq:
            ops[i].launch(False)
    ts = []
    for _ in range(reps):
        for i in seq:
            if i == target:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops[i].launch(False)
                b.record()
                ts.append((a, b))
            else:
                ops[i].launch(False)
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2]


if len(sys.argv) > 1:  # --seq 0,1: run just that sequence 40 times (for a kernel trace)
    seq = [int(x) for x in sys.argv[1].split(",")]
    for _ in range(40):
        for i in seq:
            ops[i].launch(False)
    torch.cuda.synchronize()
    sys.exit(0)

n = len(ops)
print("%-26s %8s %8s   behind: %s" % ("product", "alone", "in step", " ".join("%6d" % j for j in range(n))), flush=True)
for i in range(n):
    alone = timed([i], i)
    step = timed(list(range(n)), i)
    behind = [timed([j, i], i) if j != i else alone for j in range(n)]
    print("%d %-24s %8.4f %8.4f           %s" % (i, ops[i].name, alone, step, " ".join("%6.3f" % b for b in behind)), flush=True)
