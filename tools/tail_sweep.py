"""Where does the tail form (9 slices, the last one gathered by the launch that reduces the planes) beat the shipped
8 slices + reduce?  The kernel-choice sweep's grid, uniform graphs, sliced-eligible shapes only."""
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
DEGS = [int(v) for v in os.environ.get("DEGS", "32,64,100,200,400").split(",")]
gen = torch.Generator(device=dev).manual_seed(11)


def with_env(fn, **env):
    def run():
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return fn()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return run


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


t0 = time.time()
print("degree n_dst n_src F table_MB  sliced8_ms tail9_ms  tail/sliced")
for deg in DEGS:
    n_dst = max(64, E // deg)
    dst = torch.randint(0, n_dst, (E,), generator=gen, device=dev, dtype=torch.int32)
    for n_src in SRCS:
        src = torch.randint(0, n_src, (E,), generator=gen, device=dev, dtype=torch.int32)
        ss, ds = torch.rand(n_src, device=dev) + 0.5, torch.rand(n_dst, device=dev) + 0.5
        s8 = ops.SlicedCSR(dst, src, n_dst, n_src, n_slices=8)
        s9 = ops.SlicedCSR(dst, src, n_dst, n_src, n_slices=9)
        for F in Fs:
            if F % 4:
                continue
            X = torch.randn(n_src, F, device=dev)
            Y = torch.empty(n_dst, F, device=dev)
            t = timed({"s8": lambda: s8.spmm(X, ss, ds, out=Y),
                       "t9": with_env(lambda: s9.spmm(X, ss, ds, out=Y), DGMI_SLICED_TAIL="1", DGMI_TAIL_ROWS="2", DGMI_TAIL_PF="0")})
            print("%4d %7d %8d %4d %7.1f   %.4f %.4f   %.3f" % (deg, n_dst, n_src, F, n_src * F * 4 / 1e6, t["s8"], t["t9"], t["t9"] / t["s8"]), flush=True)
            del X, Y
        del s8, s9
        torch.cuda.empty_cache()
    print("# degree %d done, %.0f s" % (deg, time.time() - t0), flush=True)
