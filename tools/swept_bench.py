"""Experiment: row-owned slice-swept SpMM (csrc/dgmi_swept.hip) against the shipped XCD-sliced pair on the
config-4 products.  Checks the result against the sliced pair (same slice order => same in-row order when
S = 8) and against the planned kernel, then times interleaved rounds in ONE process.
  python tools/swept_bench.py [quick]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dream_gnn_amd import _lib, ops, synth

dev = torch.device("cuda:0")
L = ctypes.CDLL(_lib.LIB_PATH)
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
L.dgmi_x_spmm_swept_f32.restype = ctypes.c_int
L.dgmi_x_spmm_swept_f32.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp,
                                    ctypes.c_size_t, i32, ctypes.c_float, vp, i64, ctypes.c_float, vp]
L.dgmi_x_spmm_swept_lds_bytes.restype = ctypes.c_size_t
L.dgmi_x_spmm_swept_lds_bytes.argtypes = [i64, i32, i32, i32, i32]
N_CU = torch.cuda.get_device_properties(0).multi_processor_count


class SweptCSR:
    def __init__(self, dst, src, n_dst, n_src, F, S, vals=None, waves=16, blocks_per_cu=1):
        lpr = 8 if F <= 32 else 16 if F <= 64 else 32 if F <= 128 else 64
        self.grid, self.waves = N_CU * blocks_per_cu, waves
        tg = self.grid * waves * (64 // lpr)
        lds_cap = (160 * 1024) // blocks_per_cu
        r_need = -(-n_dst // tg)
        r_max = 31
        while r_max > 1 and L.dgmi_x_spmm_swept_lds_bytes(F, waves, r_max, S, 1) > lds_cap - 256:
            r_max -= 1
        self.Q = -(-r_need // r_max)
        self.R = -(-n_dst // (tg * self.Q))
        while L.dgmi_x_spmm_swept_lds_bytes(F, waves, self.R, S, self.Q) > lds_cap:
            self.Q += 1
            self.R = -(-n_dst // (tg * self.Q))
        self.S, self.n_dst, self.n_src, self.F, self.tg = S, n_dst, n_src, F, tg
        R, Q = self.R, self.Q
        d = dst.long()
        q = d // (tg * R)
        g = (d % (tg * R)) // R
        lr = d % R
        sw = -(-n_src // S)
        key = (((g * Q + q) * S + src.long() // sw) * R + lr).to(torch.int32)
        n_keys = tg * Q * S * R
        indptr, indices, eid = ops.csr_from_coo(key, src, n_keys)
        self.words = (indices | (lr[eid.long()].to(torch.int32) << 27)).contiguous()
        self.seg = indptr[::R].contiguous()
        assert self.seg.numel() == tg * Q * S + 1
        self.vals = None if vals is None else vals[eid.long()].contiguous()
        self.eid = eid
        self.sync = torch.zeros(8 * ((S * Q + 15) // 16 * 16), dtype=torch.int32, device=dst.device)
        self.lds = L.dgmi_x_spmm_swept_lds_bytes(F, waves, R, S, Q)

    def spmm(self, X, src_scale=None, dst_scale=None, out=None, lag=1):
        if out is None:
            out = torch.empty((self.n_dst, self.F), dtype=torch.float32, device=X.device)
        p = lambda t: None if t is None else t.data_ptr()
        st = L.dgmi_x_spmm_swept_f32(p(self.seg), p(self.words), p(self.vals), p(X), X.stride(0), p(src_scale), p(dst_scale),
                                     p(out), out.stride(0), self.n_dst, self.n_src, self.F, self.S, self.Q, self.R, self.grid,
                                     self.waves, lag, p(self.sync), self.sync.numel() * 4, 0, 0.0, None, 0, 1.0,
                                     torch.cuda.current_stream().cuda_stream)
        if st != 0:
            raise RuntimeError("dgmi_x_spmm_swept_f32 -> %d" % st)
        return out


def timeit(fns, rounds=12, inner=5):
    """Interleaved rounds: {name: median ms per call}."""
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(b) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    F = 128
    n_drug, n_dis, E = 100_000, 50_000, 10_000_000
    drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
    cases = [("drug->disease (51 MB table, 50k rows)", dis, drug, n_dis, n_drug, None, (16, 32)),
             ("disease->drug (26 MB table, 100k rows)", drug, dis, n_drug, n_dis, None, (8, 16))]
    if not quick:
        r, c, v = synth.knn_sim_graph(n_drug, 64, 2, dev)
        cases.append(("drug kNN-64 weighted (51 MB, 100k rows)", r, c, n_drug, n_drug, v, (16, 32)))
    for name, dst, src, n_dst, n_src, vals, Ss in cases:
        g = ops.CSRGraph(dst, src, n_dst, n_src, vals=vals)
        X = torch.randn(n_src, F, device=dev)
        ss = None if vals is not None else synth.degree_norm(src, n_src)
        ds = None if vals is not None else synth.degree_norm(dst, n_dst)
        y_ref = g.spmm(X, ss, ds)  # the shipped choice (sliced pair)
        y_pl = ops.spmm_csr_raw(g.indptr, g.indices, g.vals, X, ss, ds, plan=g.plan)
        scale = float(y_pl.abs().max())
        print("== %s: nnz %d, sliced-vs-planned max|d| %.2e (max|y| %.2e)" % (name, g.nnz, float((y_ref - y_pl).abs().max()), scale), flush=True)
        Y = torch.empty_like(y_ref)
        fns = {"sliced pair": lambda: g.spmm(X, ss, ds, out=Y)}
        for S in Ss:
            t0 = time.time()
            sw = SweptCSR(dst, src, n_dst, n_src, F, S, vals=vals)
            torch.cuda.synchronize()
            print("   swept S=%d: R=%d Q=%d grid=%d lds=%d B, layout %.0f ms" % (S, sw.R, sw.Q, sw.grid, sw.lds, 1e3 * (time.time() - t0)), flush=True)
            for lag in (-1, 0, 1, 2, 3):
                y = sw.spmm(X, ss, ds, lag=lag)
                err = float((y - y_pl).abs().max())
                print("      lag=%2d: max|y - planned| %.2e  %s" % (lag, err, "OK" if err <= 2e-5 * scale else "MISMATCH"), flush=True)
                assert err <= 2e-5 * scale
                y2 = sw.spmm(X, ss, ds, lag=lag)
                assert torch.equal(y, y2), "not reproducible"
                Ys = torch.empty_like(y_ref)
                fns["swept S=%d lag=%d" % (S, lag)] = (lambda sw=sw, lag=lag, Ys=Ys: sw.spmm(X, ss, ds, out=Ys, lag=lag))
        res = timeit(fns, rounds=6 if quick else 12)
        for k, v in res.items():
            print("   %-22s %.4f ms  (%.2f G edges/s)" % (k, v, g.nnz / v / 1e6), flush=True)


if __name__ == "__main__":
    main()
