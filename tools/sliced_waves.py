import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for w in (0, 32768, 40960, 54000):
        env = dict(os.environ, DGMI_SLICED_LDS=str(w))
        subprocess.run([sys.executable, __file__, str(w)], env=env, check=True)
    sys.exit(0)
import torch
from dream_gnn_amd import ops, synth
from tools.owned_bench import timeit
dev = torch.device("cuda:0")
F, ND, NS, E = 128, 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(ND, NS, E, seed=0, device=dev)
cj, ci = synth.degree_norm(drug, ND), synth.degree_norm(dis, NS)
xd, xs = torch.randn(ND, F, device=dev), torch.randn(NS, F, device=dev)
r, c, v = synth.knn_sim_graph(ND, 64, 21, dev)
line = "waves: dynamic LDS %s B per block:" % sys.argv[1]
for dst, src, n_dst, n_src, X, ss, ds, vals in ((dis, drug, NS, ND, xd, cj, ci, None), (drug, dis, ND, NS, xs, ci, cj, None), (r, c, ND, ND, xd, None, None, v)):
    sl = ops.SlicedCSR(dst, src, n_dst, n_src, vals=vals)
    out = torch.empty(n_dst, F, device=dev)
    line += "  %.4f" % timeit(lambda: sl.spmm(X, ss, ds, out=out))
print(line, flush=True)
