"""A/B in one process: the decoder's weight gradients (K = number of train pairs) as the library's single GEMM vs the
split-K batched form of model._EdgeLinear, in the lrssl-shaped and C+G-shaped training step (no augmentation): eager
wall time per step, and the same step replayed as a HIP graph (device time)."""
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
    res = {}
    for name, min_rows in (("library GEMM", 10 ** 12), ("split-K", 32768)):
        M._EdgeLinear.MIN_ROWS = min_rows
        torch.manual_seed(0)
        net = M.Net(args).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5, capturable=True)

        def step():
            net.train()
            loss, _ = H.forward_loss(net, batch, labels, 0.1)
            opt.zero_grad(set_to_none=False)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
            opt.step()
            return loss

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(5):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        eager = [wall(step) for _ in range(2)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        graph = [wall(g.replay) for _ in range(3)]
        res[name] = {"eager_ms": [round(x, 2) for x in eager], "hip_graph_replay_ms": [round(x, 2) for x in graph]}
        del g
    print(tag, res, flush=True)
