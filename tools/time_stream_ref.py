"""Reference streaming rates of this box for the narrow-design kernels' access patterns (torch kernels, not ours):
a device copy (read + write, like the solve), a sum reduction (read only, like the Hessian pass)."""
import torch
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N, D in ((4_000_000, 8), (4_000_000, 20), (4_000_000, 32), (4_000_000, 100), (1_000_000, 1000)):
    x = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    ms_c = timeit(lambda: y.copy_(x))
    ms_s = timeit(lambda: x.sum())
    print("N=%d D=%d (%.2f GB): copy %.3f ms = %.0f GB/s (read + write), sum %.3f ms = %.0f GB/s (read)"
          % (N, D, 8e-9 * N * D, ms_c, 16.0 * N * D / ms_c / 1e6, ms_s, 8.0 * N * D / ms_s / 1e6), flush=True)
    del x, y
