"""Time the batched re-weighting product W^T S (dev tool): python tools/time_reweight.py [N] [D] [K ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd import rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Ks = [int(a) for a in sys.argv[3:]] or [16, 64, 128, 256, 1024]
S = torch.randn((N, D), dtype=torch.float64, device="cuda")
for K in Ks:
    W = torch.rand((N, K), dtype=torch.float64, device="cuda")
    out = torch.empty((K, D), dtype=torch.float64, device="cuda")
    f = lambda: rows.gemm_tn(W, S, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("W^T S  N=%d D=%d K=%d: %.3f ms  %.2f TFLOP/s (2NDK)  %.0f GB/s (8N(D+K))  -> %.3g re-fits/s"
          % (N, D, K, ms, 2.0 * N * D * K / ms / 1e9, 8.0 * N * (D + K) / ms / 1e6, K / ms * 1e3))
    del W
