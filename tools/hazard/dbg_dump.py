#!/usr/bin/env python3
"""Known hazard: the fp32 merged packed-VALU kernel built with -DNMFK_DBG_DUMP=<1 H | 2 W half-step> writes, per
workgroup and lane of wave 0, the lane factor rows and the accumulated numerators it is about to finish.  ONE iteration
at rank 2, repeated beside the bf16 MFMA burner: which of them differ from the clean reference, in which lanes?
(NMFK_HIP_LIB=nmfk.jl_amd/libnmfk_hip_dbg.so; handshake with tools/hazard/dbg_dump.sh)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
from nmfk_jl_amd import _lib
lib = _lib.lib()
lib.nmfk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R, k = 4, 2
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0); ctx.set_X(X)
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)]], dtype=np.uint64)
names = ["a(e0,c0)", "a(e0,c1)", "acc(e0,c0)", "acc(e0,c1)", "a(e1,c0)", "a(e1,c1)", "acc(e1,c0)", "acc(e1,c1)", "sum x(e0)", "sum x(e1)", "k", "slot", "sum p(e0)", "sum p(e1)", "sum q(e0)", "sum q(e1)"]
def run():
    ctx.mu_sweep([k], R, seeds=seeds, maxiter=1, maxbaditers=10 ** 9)
    d = np.zeros(64 * 64 * 16, dtype=np.float32)
    rc = lib.nmfk_debug_read(d.ctypes.data, d.size)
    assert rc == 0, rc
    return d.reshape(64, 64, 16)
ref = run(); ref2 = run()
print("info", ctx.last_sweep_info(), "clean runs agree:", bool((ref.view(np.uint32) == ref2.view(np.uint32)).all()),
      "slots used:", int((ref[:, 0, 10] == 2).sum()), flush=True)
hs = os.environ.get("HANDSHAKE")
if hs:
    open(hs + ".ref", "w").close()
    while not os.path.exists(hs + ".go"):
        time.sleep(0.2)
shown = 0
tot = np.zeros(16, dtype=np.int64)
lanes = np.zeros(64, dtype=np.int64)
for rep in range(reps):
    d = run()
    diff = d.view(np.uint32) != ref.view(np.uint32)
    if not diff.any():
        continue
    tot += diff.sum((0, 1)); lanes += diff.any(2).sum(0)
    if shown < 4:
        shown += 1
        for s in range(64):
            if diff[s].any():
                fl = sorted(set(int(f) for f in np.argwhere(diff[s])[:, 1]))
                ln = sorted(set(int(l) for l in np.argwhere(diff[s])[:, 0]))
                l0 = ln[0]
                print(f"rep {rep} workgroup {s}: lanes {ln[0]}..{ln[-1]} ({len(ln)}) fields {[names[f] for f in fl]}; lane {l0}: "
                      + ", ".join(f"{names[f]} {d[s, l0, f]:.6g} (ref {ref[s, l0, f]:.6g})" for f in fl))
print("fields differing (count over reps, workgroups, lanes):", dict(zip(names, tot.tolist())))
print("lanes differing:", lanes.tolist())
