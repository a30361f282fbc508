"""Mid-range D sweep of the dense path (fused link+grad+SYRK, potrf+pack, all-rows TRSM) at a fixed N.
usage: python tools/sweep_mid.py N D [D ...]   -> one line per D (ms and TFLOP/s of useful work)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky


def timeit(fn, reps=10):
    for _ in range(10):      # the clock needs a few ms of load to settle (the first D of a sweep read 20 % slow with one call)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def one(N, D):
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)

    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    ms_h = timeit(f)
    out = f()
    H = out["H"]
    ms_p = timeit(lambda: DeviceCholesky.factor(H, check=False, pack=True))
    chol = DeviceCholesky.factor(H, pack=True)
    S = chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0)
    ms_t = timeit(lambda: chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0, out=S))
    print("N=%d D=%4d  hess %7.3f ms %5.1f TF (%4.1f%%) | potrf+pack %6.3f ms | trsm %7.3f ms %5.1f TF (%4.1f%%)" % (
        N, D, ms_h, N * D * (D + 1) / ms_h / 1e9, N * D * (D + 1) / ms_h / 1e9 / 0.786,
        ms_p, ms_t, 2.0 * N * D * D / ms_t / 1e9, 2.0 * N * D * D / ms_t / 1e9 / 0.786), flush=True)
    del X, S, obj, chol, out
    torch.cuda.empty_cache()


if __name__ == "__main__":
    N = int(sys.argv[1])
    for D in sys.argv[2:]:
        one(N, int(D))
