"""Can the plane round trip of the XCD-sliced product be hidden on CUs of its own?  Two HIP streams with
complementary CU masks (hipExtStreamCreateWithCUMask; the mask's bits are dealt round-robin over the 8 XCDs, so the
first 8 k bits are k CUs of every XCD): the sliced product on one, an HBM stream of the plane round trip's size on the
other.  Prints the product's time on fewer CUs, the stream's time on few CUs, and both together."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dream_gnn_amd import ops, synth

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(first, last):
    """stream limited to CUs [first, last) of the 256 (bit i = CU i)"""
    words = (ctypes.c_uint32 * 8)()
    for i in range(first, last):
        words[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def timed(fn, stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def together(fa, sa, fb, sb, reps=20):
    """fa on sa and fb on sb, issued alternately; ms per (fa + fb) pair by the wall clock of both streams"""
    for _ in range(3):
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
    torch.cuda.synchronize()
    a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a1, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        a0.record()
    with torch.cuda.stream(sb):
        b0.record()
    for _ in range(reps):
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
    with torch.cuda.stream(sa):
        a1.record()
    with torch.cuda.stream(sb):
        b1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) / reps, b0.elapsed_time(b1) / reps


F = 128
n_drug, n_dis, E = 100_000, 50_000, 10_000_000
drug, dis = synth.bipartite_edges(n_drug, n_dis, E, 0, dev)
cases = [("drug->disease (51 MB table, 50k rows)", dis, drug, n_dis, n_drug),
         ("disease->drug (26 MB table, 100k rows)", drug, dis, n_drug, n_dis)]
full = torch.cuda.Stream()
for name, dst, src, n_dst, n_src in cases:
    X = torch.randn(n_src, F, device=dev)
    ss = synth.degree_norm(src, n_src)
    ds = synth.degree_norm(dst, n_dst)
    sl = ops.SlicedCSR(dst, src, n_dst, n_src)
    Y = torch.empty(n_dst, F, device=dev)
    plane_bytes = 8 * n_dst * F * 4
    src_buf = torch.empty(plane_bytes // 8, dtype=torch.float32, device=dev)  # copy moves 2 x its size: half the planes each way
    dst_buf = torch.empty_like(src_buf)
    product = lambda: sl.spmm(X, ss, ds, out=Y)
    stream_op = lambda: dst_buf.copy_(src_buf)
    print("== %s; plane round trip %.0f MB, stand-in: a copy moving %.0f MB" % (name, 2 * plane_bytes / 1e6, 2 * src_buf.numel() * 4 / 1e6), flush=True)
    print("   product, all 256 CUs                  %.4f ms" % timed(product, full), flush=True)
    print("   copy,    all 256 CUs                  %.4f ms" % timed(stream_op, full), flush=True)
    for n_prod in (224, 192, 160, 128):
        sp, sc = masked_stream(0, n_prod), masked_stream(n_prod, 256)
        tp = timed(product, sp)
        tc = timed(stream_op, sc)
        ta, tb = together(product, sp, stream_op, sc)
        print("   product on %3d CUs %.4f ms | copy on %3d CUs %.4f ms | together: product stream %.4f ms, copy stream %.4f ms"
              % (n_prod, tp, 256 - n_prod, tc, ta, tb), flush=True)
