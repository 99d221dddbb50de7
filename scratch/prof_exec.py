import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
ks = list(range(2, 17))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N.execute(X, ks, R, load=False, save=False, quiet=True, seed=1, ctx=ctx, maxiter=10)
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
N.execute(X, ks, R, load=False, save=False, quiet=True, seed=2, ctx=ctx, maxiter=10)
pr.disable()
print("total", time.perf_counter() - t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
