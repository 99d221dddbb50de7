import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = np.asfortranarray(ctx.fill_uniform(1, 0, n * m).reshape(m, n).T)
ctx.set_X(X)
ks = list(range(2, 17))
N.execute(X, ks, 32, load=False, save=False, quiet=True, seed=1, ctx=ctx, maxiter=10)
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
N.execute(X, ks, 32, load=False, save=False, quiet=True, seed=2, ctx=ctx, maxiter=10)
pr.disable()
print("execute(maxiter=10):", time.perf_counter() - t, "s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
