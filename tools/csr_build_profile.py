"""f1 under rocprofv3 --kernel-trace --stats: 30 builds of the config-4 CSR (100 000 rows, 10 M edges) and nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DGMI_SKIP_BUILD", "1")
import torch
from dream_gnn_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
n_rows, n_cols, E = 100_000, 50_000, 10_000_000
row = torch.randint(0, n_rows, (E,), generator=gen, device=dev, dtype=torch.int32)
col = torch.randint(0, n_cols, (E,), generator=gen, device=dev, dtype=torch.int32)
for _ in range(30):
    out = ops.csr_from_coo(row, col, n_rows, n_cols)
torch.cuda.synchronize()
print("done")
