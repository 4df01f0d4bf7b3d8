"""(f4) the screen's workspace after one call through the C ABI: entry counts per region, flagged queries, pool use."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import _lib

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = _lib.lib
x = torch.randn(N, D, device=dev)
xn = (x / x.norm(dim=1, keepdim=True)).contiguous()
need = lib.dgmi_knn_cosine_workspace_bytes(N, D, k)
ws = torch.zeros(need, dtype=torch.uint8, device=dev)
nbr = torch.empty(N, k, dtype=torch.int32, device=dev)
rc = lib.dgmi_knn_cosine_topk_f32(ctypes.c_void_p(xn.data_ptr()), D, N, D, k, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(ws.data_ptr()), need, None)
torch.cuda.synchronize()
print("rc", rc, "workspace MB", need / 1e6)
al = lambda b: (b + 255) & ~255
big = N >= 49152
tile = 256 if big else 128
Np = (N + tile - 1) // tile * tile
Dp = (D + 63) // 64 * 64
cap_r = 64 if k <= 8 else (128 if k <= 16 else 256)
sym = N >= (24576 if k <= 8 else 40960)
cap_c = 8 * cap_r if sym else 0
at = 0
xb = at; at += al(Np * Dp * 2)
part = at; at += al(N * 64 * 4 * 4)
tau = at; at += al(N * 4)
thr = at; at += al(Np * 4)
cnt = at; at += al(N * 9 * 4)
buf = at; at += al(N * (8 * cap_r + cap_c) * 8)
flags = at; at += al(N * 4)
chunks = (N * (16 * k + 48) + 255) // 256 + 2048 if big else 0
pool = at; at += al(chunks * 256 * 16)
ctl = at; at += al((chunks + 1) * 4 if chunks else 0)
print("layout total", at, "==", need)
c = ws[cnt:cnt + N * 9 * 4].view(torch.int32).view(N, 9).cpu()
f = ws[flags:flags + N * 4].view(torch.int32).cpu()
print("flagged queries", int(f.sum()), "of", N)
print("entries per query: rows", float(c[:, :8].clamp(max=100000).sum(1).float().mean()), "column", float(c[:, 8].clamp(max=100000).float().mean()))
print("max per region", c[:, :8].max().item(), "max column", c[:, 8].max().item(), "marks", int((c >= 0x40000000).sum()))
if chunks:
    pc = ws[ctl:ctl + (chunks + 1) * 4].view(torch.int32).cpu()
    used = pc[1:]
    print("chunks handed out", int(pc[0]), "of", chunks, "records", int(used.sum()), "chunks with records", int((used > 0).sum()))
    t = ws[thr:thr + Np * 4].view(torch.float32).cpu()
    print("thr min/mean/max", float(t[:N].min()), float(t[:N].mean()), float(t[:N].max()), "pad", t[N:].unique())
