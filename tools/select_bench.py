"""Device time of the per-step edge-dropout subset selections at the reference's dataset sizes (lrssl shape:
relations of 464 897 / 2 746 edges; kNN-4 graphs of ~6-7 k entries) and at config 4's 10 M-edge lists, with the
single-workgroup / window-passes line at 2^16 (shipped) and at 2^20 (round 2), then the lrssl-shaped training
step with either (in one process, interleaved)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from dream_gnn_amd import ops
from dream_gnn_amd import _lib  # dgmi_set_tuning
dev = torch.device("cuda:0")
def t(fn, reps=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for wmin in ("65536", "1048576"):
    _lib.set_tuning("select_window_min", int(wmin))
    for name, Es in (("lrssl encoder graph: 4 relations", [464897, 2746, 464897, 2746]), ("lrssl kNN-4 adjacencies", [6825, 6109, 6800, 6100]),
                     ("C+G merged encoder graph", [407080, 4019, 407080, 4019]), ("one 65 535-edge list", [65535]), ("one 65 536-edge list", [65536]),
                     ("config 4: one 10 M-edge list", [10_000_000])):
        keeps = [max(1, int(e * 0.9)) for e in Es]
        us = t(lambda: ops.random_subset_select_batch(Es, keeps, list(range(1, len(Es) + 1)), dev))
        print("window from %7s edges | %-36s %8.1f us per batch" % (wmin, name, us), flush=True)
spec = importlib.util.spec_from_file_location("msb", os.path.join(os.path.dirname(__file__), "model_step_bench.py"))
src = open(spec.origin).read().split("if os.environ.get(\"ONLY\")")[0]
ns_ = {"__file__": spec.origin}
exec(compile(src, "msb", "exec"), ns_)
H, M = ns_["H"], ns_["M"]
batch, labels, args = ns_["problem"](763, 681, 768, 128)
torch.manual_seed(0)
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
res = {}
for rnd in range(4):
    for wmin in ("65536", "1048576"):
        _lib.set_tuning("select_window_min", int(wmin))
        for _ in range(5): H.train_step(net, opt, batch, labels, do_augment=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): H.train_step(net, opt, batch, labels, do_augment=True)
        torch.cuda.synchronize()
        res.setdefault(wmin, []).append((time.perf_counter() - t0) / 30 * 1e3)
for wmin, v in res.items():
    print("lrssl-shaped training step, window from %7s edges: min %.3f ms  (%s)" % (wmin, min(v), " ".join("%.3f" % x for x in v)), flush=True)
