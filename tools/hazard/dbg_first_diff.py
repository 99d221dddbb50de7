#!/usr/bin/env python3
"""Known hazard: ONE iteration of the fp32 merged packed-VALU kernel, repeated; where exactly does a repetition differ
from the first?  (run beside tools/hazard/burner 0 in another process: tools/hazard/dbg_first_diff.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(os.environ.get("ITERS", 1))
n, m = (int(os.environ.get("N", 700)), int(os.environ.get("M", 130)))
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0)
ctx.set_X(X)
ks = [int(v) for v in os.environ.get("KS", "2").split(",")]
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ref, nbad, shown = None, 0, 0
for rep in range(reps):
    res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9,
                       **({"compute": NMFk.COMPUTE_F64} if os.environ.get("COMPUTE") == "f64" else {}))
    if ref is None:
        ref = res; print(ctx.last_sweep_info(), flush=True)
        if os.environ.get("HANDSHAKE"):  # the reference is computed with the GPU to ourselves; the shell starts the burner now
            import time
            open(os.environ["HANDSHAKE"] + ".ref", "w").close()
            while not os.path.exists(os.environ["HANDSHAKE"] + ".go"):
                time.sleep(0.2)
        continue
    bad = False
    for k in ks:
        for name in ("W", "H"):
            a, b = res[k][name], ref[k][name]
            if (a == b).all():
                continue
            bad = True
            if shown < int(os.environ.get("SHOW", 12)):
                for r in range(R):
                    d = np.argwhere(a[r] != b[r])
                    if len(d) == 0:
                        continue
                    rel = np.abs(a[r] - b[r])[a[r] != b[r]] / np.abs(b[r]).max()
                    if name == "W":
                        rows, sig = sorted(set(int(v) for v in d[:, 0])), sorted(set(int(v) for v in d[:, 1]))
                    else:
                        rows, sig = sorted(set(int(v) for v in d[:, 1])), sorted(set(int(v) for v in d[:, 0]))
                    if name == "W" and len(rows) < n:  # were these rows simply NOT updated?  (W is updated in place; the
                        # final scaling multiplies every row by the row sums of H: old row * t, the same t for every row)
                        W0, _ = oracle.init_factors(int(seeds[ks.index(k), r]), n, m, k)
                        t = a[r][rows, :] / W0[rows, :]
                        print(f"   corrupted rows / initial rows: per-signal spread over the rows {t.std(0) / t.mean(0)} (0 = the rows were not updated)")
                    print(f"rep {rep} k {k} {name} restart {r}: {len(d)} entries, lane elements {rows[:20]}{'...' if len(rows) > 20 else ''} "
                          f"signals {sig}, rel diff {rel.min():.2e}..{rel.max():.2e}")
                shown += 1
    nbad += bad
print("reps", reps - 1, "differing", nbad)
