"""A/B in one process: lrssl-shaped full training step and eval forward with the complement form of the
label-0 relation on / off (GCMCLayer.complement_form), eager and as a replayed HIP graph (where the host
no longer hides the device time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
spec = importlib.util.spec_from_file_location("msb", os.path.join(os.path.dirname(__file__), "model_step_bench.py"))
src = open(spec.origin).read().split("if os.environ.get(\"ONLY\")")[0]
ns_ = {"__file__": spec.origin}
exec(compile(src, "msb", "exec"), ns_)
H, M, L, dev = ns_["H"], ns_["M"], ns_["L"], ns_["dev"]

for tag, nd, ns, out_units in (("cfg2 lrssl-shape", 763, 681, 128), ("cfg3 C+G-shape (one block)", 1256, 722, 256)):
    batch, labels, args = ns_["problem"](nd, ns, 768, out_units)
    torch.manual_seed(0)
    net = M.Net(args).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
    res = {}
    for rnd in range(3):
        for comp in (False, True):
            L.GCMCLayer.complement_form = comp
            for aug in (False, True):
                for _ in range(5): H.train_step(net, opt, batch, labels, do_augment=aug)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30): H.train_step(net, opt, batch, labels, do_augment=aug)
                torch.cuda.synchronize()
                res.setdefault((comp, "train step, augment=%s" % aug), []).append((time.perf_counter() - t0) / 30 * 1e3)
            net.eval()
            with torch.no_grad():
                for _ in range(3): H.forward_loss(net, batch, labels, 0.1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30): H.forward_loss(net, batch, labels, 0.1)
                torch.cuda.synchronize()
                res.setdefault((comp, "eval forward"), []).append((time.perf_counter() - t0) / 30 * 1e3)
            net.train()
    for what in ("train step, augment=False", "train step, augment=True", "eval forward"):
        a, b = min(res[(False, what)]), min(res[(True, what)])
        print("%s | %-26s plain %.3f ms   complement %.3f ms   (%+.1f %%)" % (tag, what, a, b, (b / a - 1) * 100), flush=True)
    # device time of the eval forward: replay a captured graph (no host gaps)
    net.eval()
    for comp in (False, True):
        L.GCMCLayer.complement_form = comp
        with torch.no_grad():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): H.forward_loss(net, batch, labels, 0.1)
            torch.cuda.current_stream().wait_stream(s)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    H.forward_loss(net, batch, labels, 0.1)
                for _ in range(5): g.replay()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(50): g.replay()
                torch.cuda.synchronize()
                print("%s | eval forward as a replayed HIP graph, complement=%s: %.3f ms" % (tag, comp, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
            except Exception as exc:  # noqa: BLE001
                print("graph capture failed:", repr(exc)[:200], flush=True)
