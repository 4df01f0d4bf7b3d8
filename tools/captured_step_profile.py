import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
sys.argv = [sys.argv[0]]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(root, "tools", "model_step_bench.py")).read().split("if os.environ.get(\"ONLY\")")[0]
ns = {"__file__": os.path.join(root, "tools", "model_step_bench.py")}
exec(compile(src, "msb", "exec"), ns)
H, M, dev = ns["H"], ns["M"], ns["dev"]
batch, labels, args = ns["problem"](763, 681, 768, 128)
torch.manual_seed(0)
net = M.Net(args).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-5, capturable=True)
step = H.CapturedTrainStep(net, opt, batch, labels)
for _ in range(50):
    step()
torch.cuda.synchronize()
