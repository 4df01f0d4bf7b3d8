"""Does a whole training step through the HIP path capture into a HIP graph (torch.cuda.graphs)?
lrssl-shaped model, no augmentation inside the captured region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("msb", os.path.join(os.path.dirname(__file__), "model_step_bench.py"))
src = open(spec.origin).read().split("if os.environ.get(\"ONLY\")")[0]
ns = {"__file__": spec.origin}
exec(compile(src, "msb", "exec"), ns)
H, M, dev = ns["H"], ns["M"], ns["dev"]
batch, labels, args = ns["problem"](763, 681, 768, 128)
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
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 30
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 30
print(f"eager {eager*1e3:.2f} ms/step, HIP-graph replay {graph*1e3:.2f} ms/step, loss {float(static_loss):.4f}")
