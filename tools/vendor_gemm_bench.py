"""vendor-library yardstick: torch (hipBLASLt / rocBLAS) bf16 NT GEMM on the step's shapes, cold operands (rotating buffers
larger than the 256 MB Infinity Cache), HIP-event timing.  Not part of the product path -- a ceiling estimate for DESIGN.md."""
import sys, torch
torch.manual_seed(0)
dev = "cuda"
shapes = [(12608, 3072, 768), (12608, 768, 3072), (12608, 2304, 768), (12608, 768, 768),
          (8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768), (8192, 768, 2304)]
print("%7s %6s %6s %9s %9s" % ("M", "N", "K", "avg_us", "TFLOP/s"))
for M, N, K in shapes:
    nbuf = max(2, int(600e6 // ((M * K + N * K + M * N) * 2)) + 1)
    A = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    W = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(nbuf)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    for i in range(nbuf * 2):
        torch.matmul(A[i % nbuf], W[i % nbuf].t(), out=C[i % nbuf])
    torch.cuda.synchronize()
    reps = 5 * nbuf
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        evs[i][0].record()
        torch.matmul(A[i % nbuf], W[i % nbuf].t(), out=C[i % nbuf])
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    t = sum(ts) / len(ts)
    print("%7d %6d %6d %9.2f %9.1f   (median %.2f us)" % (M, N, K, t * 1e3, 2.0 * M * N * K / (t * 1e-3) * 1e-12, ts[len(ts) // 2] * 1e3), flush=True)
