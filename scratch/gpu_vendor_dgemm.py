"""Scratch: what the vendor FP64 GEMM reaches on the shapes of the trailing update (context for
DESIGN §5; not part of the product)."""
import torch, time
torch.manual_seed(0)
dev = "cuda"
for (m, n, k) in [(8192, 8192, 512), (16384, 16384, 512), (32768, 32768, 512), (16384, 16384, 256), (8192, 8192, 8192), (30000, 30000, 512)]:
    a = torch.randn(m, k, dtype=torch.float64, device=dev)
    b = torch.randn(n, k, dtype=torch.float64, device=dev)
    c = torch.randn(m, n, dtype=torch.float64, device=dev)
    for _ in range(2):
        c.addmm_(a, b.t(), beta=1.0, alpha=-1.0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        c.addmm_(a, b.t(), beta=1.0, alpha=-1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("C[%d,%d] -= A[%d,%d] B^T : %.3f ms  %.1f TFLOP/s" % (m, n, m, k, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
    del a, b, c
