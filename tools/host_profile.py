"""Where the HOST time of an eager lrssl-shaped training step goes (cProfile, cumulative, this repo's functions)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = [sys.argv[0]]
src = open(os.path.join(os.path.dirname(__file__), "model_step_bench.py")).read().split("if os.environ.get(\"ONLY\")")[0]
ns = {"__file__": os.path.join(os.path.dirname(__file__), "model_step_bench.py")}
exec(compile(src, "msb", "exec"), ns)
H, M, dev = ns["H"], ns["M"], ns["dev"]
batch, labels, args = ns["problem"](763, 681, 768, 128)
torch.manual_seed(0)
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5)
for _ in range(10):
    H.train_step(net, opt, batch, labels)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    H.train_step(net, opt, batch, labels)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
st.print_stats(70)
out = s.getvalue()
print("\n".join(l for l in out.splitlines() if "dream_gnn_amd" in l or "tottime" in l or "function calls" in l or "built-in" in l or "{method" in l)[:9000])
