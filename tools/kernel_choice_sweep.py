"""Is the kernel form `ops.CSRGraph.spmm` picks the fastest one away from the tuned shapes?  

Sweep: F in {64, 128, 256, 344}, n_src 20k .. 1.6M, average in-degree 16 .. 400, uniform and Zipf(1.1)
destination degrees, 6.4 M edges each.  For every graph every eligible form is timed (interleaved, same
process): wave-per-row, planned, XCD-sliced (rule's column passes / full width), split-sliced for the skewed
graphs; then the form the rule picks.  A pick more than 10 % slower than the best form is a MIS-PICK.
Writes gpurun_out/kernel_choice_sweep.csv (copied to profiles/ by hand) and prints the mis-picks."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dream_gnn_amd import ops

dev = torch.device("cuda:0")
E = int(os.environ.get("EDGES", 6_400_000))
Fs = [int(v) for v in os.environ.get("FS", "64,128,256,344").split(",")]
SRCS = [int(v) for v in os.environ.get("SRCS", "20000,50000,100000,200000,400000,800000,1600000").split(",")]
DEGS = [int(v) for v in os.environ.get("DEGS", "16,32,64,100,200,400").split(",")]
gen = torch.Generator(device=dev).manual_seed(11)
cpu_gen = torch.Generator().manual_seed(12)


def timed(fns, rounds=4, inner=3):
    for f in fns.values():
        f()
        f()
    torch.cuda.synchronize()
    best = {k: float("inf") for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], a.elapsed_time(b) / inner)
    return best


rows, mispicks = [], []
t_start = time.time()
for skew in (False, True):
    for deg in DEGS:
        n_dst = max(64, E // deg)
        if skew:
            p = 1.0 / torch.arange(1, n_dst + 1, dtype=torch.float64) ** 1.1
            dst = torch.multinomial(p / p.sum(), E, replacement=True, generator=cpu_gen).to(torch.int32).to(dev)
        else:
            dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
        for n_src in SRCS:
            src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
            g = ops.CSRGraph(dst, src, n_dst, n_src)
            S = g._S
            ss, ds = torch.rand(n_src, device=dev) + 0.5, torch.rand(n_dst, device=dev) + 0.5
            sliced = split = None
            for F in Fs:
                X = torch.randn(n_src, F, device=dev)
                Y = torch.empty(n_dst, F, device=dev)
                forms = {"rowwave": lambda: ops.spmm_csr_raw(S.indptr, S.indices, None, X, ss, ds, out=Y),
                         "planned": lambda: ops.spmm_csr_raw(S.indptr, S.indices, None, X, ss, ds, out=Y, plan=S.plan)}
                if skew and n_dst < 50_000:
                    del forms["rowwave"]  # one wave per multi-million-edge row: tens of ms, and never what a rule would pick
                if F % 4 == 0 and n_dst * 8 < 2 ** 31 - 1:
                    if S.regular:
                        if sliced is None:
                            sliced = ops.SlicedCSR.from_csr(S.indptr, S.indices, S.eid, n_dst, n_src)
                        forms["sliced"] = lambda: sliced.spmm(X, ss, ds, out=Y)
                        forms["sliced_fullwidth"] = lambda: sliced.spmm(X, ss, ds, out=Y, full_width=True)
                    else:
                        if split is None:
                            split = ops._SplitSliced(S.indptr, S.eid, S.src, n_dst, n_src)
                        forms["split_sliced"] = lambda: split.spmm(X, ss, ds, Y, None)
                picked = ("sliced" if g._use_sliced(F, n_dst, n_src, S.regular) and F % 4 == 0 else
                          "split_sliced" if g._use_split(F, n_dst, n_src, S.regular) and F % 4 == 0 else "planned")
                t = timed(forms)
                t_pick = timed({"pick": lambda: g.spmm(X, ss, ds, out=Y)})["pick"]
                best = min(t, key=t.get)
                row = {"skew": "zipf1.1" if skew else "uniform", "avg_degree": deg, "n_dst": n_dst, "n_src": n_src, "F": F,
                       "table_MB": round(n_src * F * 4 / 1e6, 1), "regular": bool(S.regular), "picked": picked,
                       "picked_ms": round(t_pick, 4), "best": best, "best_ms": round(t[best], 4),
                       "loss_pct": round((t_pick / t[best] - 1) * 100, 1)}
                for k in ("rowwave", "planned", "sliced", "sliced_fullwidth", "split_sliced"):
                    row[k + "_ms"] = round(t[k], 4) if k in t else ""
                rows.append(row)
                if t_pick > 1.10 * t[best]:
                    mispicks.append(row)
                del X, Y
            del g, sliced, split
            torch.cuda.empty_cache()
        print("skew=%s degree=%d done (%d rows, %d mis-picks, %.0f s)" % (skew, deg, len(rows), len(mispicks), time.time() - t_start), flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_choice_sweep.csv")
with open(out, "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
    wr.writeheader()
    wr.writerows(rows)
print("%d shapes, %d mis-picks (> 10 %% slower than the best form)" % (len(rows), len(mispicks)))
for r in sorted(mispicks, key=lambda r: -r["loss_pct"])[:60]:
    print({k: r[k] for k in ("skew", "avg_degree", "n_dst", "n_src", "F", "table_MB", "picked", "picked_ms", "best", "best_ms", "loss_pct")})
