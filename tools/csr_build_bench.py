"""f1: device COO -> CSR and the two sliced builders on the hand-written record sort (dgmi_sort.hip): arrays checked
against torch's stable sort of the same keys (bit-identical), device time per build.  The library-sort numbers this
replaced are in profiles/r03_csr_build/ (measured in the same process while both paths existed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dream_gnn_amd import _lib, ops
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)

def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for n_rows, n_cols, E in ((50_000, 100_000, 10_000_000), (100_000, 50_000, 10_000_000), (100_000, 100_000, 12_900_000), (800_000, 400_000, 10_000_000),
                          (681, 763, 467_643), (763, 763, 6_825), (300, 200, 1), (5, 7, 0), (70_000, 70_000, 8191), (70_000, 70_000, 8193), (2_000_000, 10, 3_000_000)):
    row = torch.randint(0, n_rows, (E,), generator=gen, device=dev, dtype=torch.int32)
    col = torch.randint(0, n_cols, (E,), generator=gen, device=dev, dtype=torch.int32)
    own = ops.csr_from_coo(row, col, n_rows, n_cols, return_flag=True)
    t_own = t(lambda: ops.csr_from_coo(row, col, n_rows, n_cols))
    _lib.set_tuning("sort_plain_tiles", 1)  # A/B: tile = blockIdx.x (rounds 1-3) against the XCD-aware tile order
    plain = ops.csr_from_coo(row, col, n_rows, n_cols)
    t_plain = t(lambda: ops.csr_from_coo(row, col, n_rows, n_cols))
    _lib.set_tuning("sort_plain_tiles", 0)
    assert all(torch.equal(a, b) for a, b in zip(own[:3], plain))
    order = torch.sort(row.long(), stable=True).indices
    ref_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    ref_ptr[1:] = torch.cumsum(torch.bincount(row.long(), minlength=n_rows), 0)
    same = torch.equal(own[0].long(), ref_ptr) and torch.equal(own[1], col[order]) and torch.equal(own[2].long(), order)
    line = "%9d rows %9d edges: CSR %.3f ms (plain tile order %.3f)" % (n_rows, E, t_own, t_plain)
    if n_rows * 8 < 2**31 and E > 0:
        sl = ops.SlicedCSR(row, col, n_rows, n_cols)
        sl2 = ops.SlicedCSR.from_csr(own[0], own[1], own[2], n_rows, n_cols)
        same &= all(torch.equal(a, b) for a, b in zip((sl.segptr, sl.indices, sl.eid), (sl2.segptr, sl2.indices, sl2.eid)))
        same &= torch.equal(col[sl.eid.long()], sl.indices)
        # inside every (slice, row) segment edge ids ascend and the rows match
        seg = torch.repeat_interleave(torch.arange(8 * n_rows, device=dev), (sl.segptr[1:] - sl.segptr[:-1]).long())
        same &= torch.equal(seg % n_rows, row[sl.eid.long()].long())
        d = sl.eid[1:].long() - sl.eid[:-1].long()
        same &= bool(((d > 0) | (seg[1:] != seg[:-1])).all())
        t_coo = t(lambda: ops.SlicedCSR(row, col, n_rows, n_cols))
        t_csr = t(lambda: ops.SlicedCSR.from_csr(own[0], own[1], own[2], n_rows, n_cols))
        line += "   sliced from COO %.3f ms   sliced from the CSR %.3f ms" % (t_coo, t_csr)
    print(line + "   identical: %s" % same, flush=True)
    assert same
# an out-of-range id: flag set, nothing written out of bounds
row = torch.randint(0, 1000, (100_000,), generator=gen, device=dev, dtype=torch.int32)
row[777] = 5000
col = torch.randint(0, 50, (100_000,), generator=gen, device=dev, dtype=torch.int32)
*_, flag = ops.csr_from_coo(row, col, 1000, 50, return_flag=True)
print("range flag on a bad id:", int(flag.item()))
assert int(flag.item()) == 1
