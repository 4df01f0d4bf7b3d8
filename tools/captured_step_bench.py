"""The real-dataset-shaped training step WITH the reference's per-step augmentation (train.py:267: edge dropout on 4
relations + 4 similarity graphs, feature noise): eager (`harness.train_step`) against the same step recorded once as a
HIP graph (`harness.CapturedTrainStep`, subsets' seeds drawn on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = [sys.argv[0]]
src = open(os.path.join(os.path.dirname(__file__), "model_step_bench.py")).read().split("if os.environ.get(\"ONLY\")")[0]
ns = {"__file__": os.path.join(os.path.dirname(__file__), "model_step_bench.py")}
exec(compile(src, "msb", "exec"), ns)
H, M, dev = ns["H"], ns["M"], ns["dev"]


def wall(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for tag, nd, nsz, out in (("cfg2 lrssl-shape", 763, 681, 128), ("cfg3 C+G-shape", 1256, 722, 256)):
    batch, labels, args = ns["problem"](nd, nsz, 768, out)
    torch.manual_seed(0)
    net = M.Net(args).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
    eager = [wall(lambda: H.train_step(net, opt, batch, labels, do_augment=True)) for _ in range(2)]
    torch.manual_seed(0)
    net = M.Net(args).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5, capturable=True)
    step = H.CapturedTrainStep(net, opt, batch, labels)
    losses = [float(step()) for _ in range(5)]
    graph = [wall(step) for _ in range(3)]
    print(tag, "eager ms/step", [round(x, 2) for x in eager], "| captured ms/step", [round(x, 2) for x in graph],
          "| first losses", [round(x, 4) for x in losses], flush=True)
