"""planted rank-6 8192x512 matrix, execute(X, 2:16, 32) in fp32 and fp64 compute modes.
Prints kopt, robustness and fit per k, and the time of each sweep (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as NMFk

def planted(ctx, n=8192, m=512, k0=6, seed=2):
    W0 = ctx.fill_uniform(seed, 0, n * k0).reshape(k0, n).T.astype(np.float64)
    H0 = ctx.fill_uniform(seed, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    U = ctx.fill_uniform(seed, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
    return np.asfortranarray((W0 @ H0 + 0.01 * U).astype(np.float32))

if __name__ == "__main__":
    ctx = NMFk.Context(0)
    X = planted(ctx)
    ctx.set_X(X)
    modes = sys.argv[1:] or ["f32", "f64"]
    for mode in modes:
        t = time.perf_counter()
        W, H, fit, rob, aic, kopt, det = NMFk.execute(X, range(2, 17), 32, load=False, save=False, quiet=True, seed=2, ctx=ctx,
                                                      compute=mode, return_details=True)
        dt = time.perf_counter() - t
        print(mode, "kopt", kopt, "time %.1f s" % dt, "schedule", ctx.last_sweep_info())
        for k in range(2, 17):
            print("  k=%2d fit %10.5f rob %8.4f mean iters %7.1f" % (k, fit[k - 1], rob[k - 1], det[k]["iters"].mean()))
        sys.stdout.flush()
