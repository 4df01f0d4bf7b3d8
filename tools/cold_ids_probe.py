"""Is the in-step slow-down of a sliced product the cold id stream?  Product 1 of bench.py's step (26 MB table, 100 k
destination rows) in a loop of its own, with the memory-side cache flushed in between, and with single inputs
re-warmed after the flush."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
ops, _, _ = bench.build_ops(torch, 0, 1, dev, "nodes")
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
op = ops[which]
sl = op.shard.local._S.sliced
junk = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
sink = torch.zeros(1, dtype=torch.int64, device=dev)


def flush():
    junk.add_(1)  # 768 MB read + written: nothing of the product is left in the 256 MB Infinity Cache


def touch(*ts):
    for t in ts:
        sink.add_(t.view(-1)[: t.numel() // 2 * 2].view(torch.int64).sum() if t.dtype != torch.float32 else t.sum().long())


def med(pre, reps=25):
    ts = []
    for _ in range(reps + 3):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        op.launch(False)
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts[3:])
    return v[len(v) // 2]


from dream_gnn_amd import _lib  # launch-parameter overrides: dgmi_set_tuning (the library reads no env per launch)

print("product:", op.name)
_lib.set_tuning("sliced_touch_lead", 0)
print("touch-ahead off:")
print("  loop of its own                           %.4f ms" % med(lambda: None))
print("  flushed before every call                 %.4f ms" % med(flush))
print("  flushed, then ids re-read                 %.4f ms" % med(lambda: (flush(), touch(sl.indices))))
print("  flushed, then segptr re-read              %.4f ms" % med(lambda: (flush(), touch(sl.segptr))))
print("  flushed, then ids + segptr re-read        %.4f ms" % med(lambda: (flush(), touch(sl.indices, sl.segptr))))
print("  flushed, then the table re-read           %.4f ms" % med(lambda: (flush(), touch(op.X))))
_lib.set_tuning("sliced_touch_lead", 24)
print("touch-ahead 24:")
print("  loop of its own                           %.4f ms" % med(lambda: None))
print("  flushed before every call                 %.4f ms" % med(flush))
print("  flushed, then ids re-read                 %.4f ms" % med(lambda: (flush(), touch(sl.indices))))
print("  flushed, then segptr re-read              %.4f ms" % med(lambda: (flush(), touch(sl.segptr))))
print("  flushed, then ids + segptr re-read        %.4f ms" % med(lambda: (flush(), touch(sl.indices, sl.segptr))))
print("  flushed, then the table re-read           %.4f ms" % med(lambda: (flush(), touch(op.X))))
for d in (0, 4, 8, 16, 24, 32, 48, 64, 96, 128, 256):
    _lib.set_tuning("sliced_touch_lead", d)
    print("touch-ahead %3d blocks: loop of its own %.4f ms   flushed before every call %.4f ms" % (d, med(lambda: None), med(flush)), flush=True)
