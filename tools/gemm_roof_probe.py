"""What a library bf16 GEMM reaches on the kNN screen's shape (K = 768: short) — the practical MFMA roof the screen
kernel of DESIGN 4.7 can be compared with (it additionally thresholds and appends every score)."""
import torch
dev = torch.device("cuda:0")
D = 768
for M, N in ((8192, 100_096), (16384, 100_096), (32768, 32768), (8192, 8192)):
    a = torch.randn(M, D, device=dev).bfloat16()
    b = torch.randn(N, D, device=dev).bfloat16()
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        c = a @ b.t()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("bf16 %6d x %6d x %d (NT, bf16 out): %.3f ms = %.0f TFLOP/s" % (M, N, D, ms, 2.0 * M * N * D / ms / 1e9), flush=True)
    del c
    af, bf = a.float(), b.float()
