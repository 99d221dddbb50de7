import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import nmfk_jl_amd as N
import nmfk_oracle as o
ctx = N.Context(0)
n, m = 40, 33
X = o.uniform_fill(11, 0, n*m).reshape(n, m).astype(np.float32)
ctx.set_X(X)
for k in (2, 1):
    for comp in (0, 1):
        for it in (5, 10, 60):
            print('k', k, 'compute', comp, 'maxiter', it, flush=True)
            seeds = np.array([[N.run_seed(5, k, r) for r in range(3)]], dtype=np.uint64)
            t = time.time()
            res = ctx.mu_sweep([k], 3, seeds=seeds, maxiter=it, compute=comp, maxbaditers=10**9)[k]
            print('   gpu done', time.time() - t, res['iters'], res['objvalue'], flush=True)
            W0, H0 = o.init_factors(int(seeds[0, 0]), n, m, k)
            t = time.time()
            ref = o.singlerun(X, k, W0, H0, maxiter=it, maxbaditers=10**9)
            print('   oracle done', time.time() - t, ref['objvalue'], flush=True)
