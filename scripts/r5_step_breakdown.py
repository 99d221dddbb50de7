"""Where a bench step's time goes OUTSIDE the MU loop (round 5): NMFk.execute on the headline workload (8192 x 512, k = 2:16, 32 restarts)
with a short MU budget (maxiter=200), under cProfile.  usage: r5_step_breakdown.py [maxiter]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as NMFk

maxiter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, m, ks, R = 8192, 512, list(range(2, 17)), 32
ctx = NMFk.Context(0)
X = np.random.default_rng(1).random((n, m), dtype=np.float32)
ctx.set_X(X)
kw = dict(load=False, save=False, quiet=True, seed=1, ctx=ctx, maxiter=maxiter)
NMFk.execute(X, ks, R, **kw)  # warm-up
for rep in range(2):
    ctx.set_profiling(True)
    t = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    NMFk.execute(X, ks, R, **kw)
    pr.disable()
    wall = time.perf_counter() - t
    prof = ctx.get_profile()
    ctx.set_profiling(False)
    loop = prof.get("mu_loop", {}).get("ms", 0.0)
    print(f"rep {rep}: execute {wall * 1e3:.1f} ms, MU loop (GPU) {loop:.1f} ms, rest {wall * 1e3 - loop:.1f} ms")
    for name, e in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]:
        print(f"   {name:28s} {e['ms']:9.2f} ms  {e['launches']:6d} launches")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
