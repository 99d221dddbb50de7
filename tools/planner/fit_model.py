#!/usr/bin/env python3
"""Calibration of the launch-geometry cost model of nmfk_mu_sweep (nmfk_api.hip: hyb_stream_cost / hyb_res_cost) against launch
durations measured on MI355X (profiles/r05/geometry_scan.txt: HIP-event samples of single launches, 8192 x 512, ranks 2..16 in
equal numbers).  A Python replica of the model's event simulation; grid search over its constants.  Not part of the product."""
import heapq, itertools, sys
CUS = 256
RATIO = {16: 1.0, 8: 0.76, 4: 0.52}

def mix(units):  # ranks 2..16 in equal numbers: 8 : 4 : 3 of the variants 16 : 8 : 4, list order 16, 8, 4
    n16 = units * 8 // 15; n8 = units * 4 // 15; n4 = units - n16 - n8
    return [16] * n16 + [8] * n8 + [4] * n4

def cu_share(t, wpu, wpc, a):
    """t[u]: us of unit u's workgroups on a full CU; rate(res) = 1 / (a + (1 - a) res / wpc)"""
    rate = lambda r: 1.0 / (a + (1 - a) * r / wpc)
    wl = [x for x in t for _ in range(wpu)]
    it = iter(wl)
    cu = [[] for _ in range(CUS)]
    for slot in range(wpc):
        for c in range(CUS):
            w = next(it, None)
            if w is None: break
            cu[c].append(w)
    last = [0.0] * CUS
    heap = [(min(q) / rate(len(q)), c) for c, q in enumerate(cu) if q]
    heapq.heapify(heap)
    ms = 0.0
    while heap:
        tm, c = heapq.heappop(heap)
        q = cu[c]
        done = (tm - last[c]) * rate(len(q))
        q[:] = [r - done for r in q if r - done > 1e-9]
        last[c] = tm; ms = max(ms, tm)
        while len(q) < wpc:
            w = next(it, None)
            if w is None: break
            q.append(w)
        if q: heapq.heappush(heap, (tm + min(q) / rate(len(q)), c))
    return ms

def stream(units, L, D, ws, S, P):
    fix, c16, a, pw, launch = P
    lt = 32 if ws > 1 else 256
    ntile = -(-L // lt)
    dchunk = -(-D // S)
    if S > 1: dchunk = (dchunk + 15) & ~15
    nch = ((-(-dchunk // ws)) + 15) >> 4 if ws > 1 else (dchunk + 15) >> 4
    t = [fix + nch * c16 * RATIO[v] * (pw if ws > 1 else 1.0) for v in mix(units)]
    return launch + cu_share(t, ntile * S, 4 if ws == 4 else 2, a)

def listsched(t, wpu, slots):
    wl = [x for x in t for _ in range(wpu)]
    if len(wl) <= slots: return max(wl)
    h = [0.0] * slots
    for w in wl:
        heapq.heappush(h, heapq.heappop(h) + w)
    return max(h)

def res(units, L, D, g, P):
    st0, st1, c16, pfix, launch = P
    ntp = -(-L // 32); nch = ((D + 63) & ~63) >> 4
    pairs = -(-ntp // (16 * g))
    t = [st0 + st1 * (D / 512.0) * (v / 16.0) + pairs * (nch * c16 * RATIO[v] + pfix) for v in mix(units)]
    return launch + listsched(t, g, CUS)

# measured kernel durations (us), H half-step streaming at 8192 x 512 (L = 512, D = 8192): (units, ws, S): us
H = {(60, 1, 1): 230, (60, 1, 2): 124, (60, 1, 3): 137, (60, 1, 4): 109, (60, 1, 5): 115, (60, 1, 6): 107, (60, 1, 8): 104, (60, 1, 12): 112,
     (60, 1, 16): 112, (60, 8, 1): 129, (60, 8, 2): 128, (60, 4, 1): 127, (60, 4, 2): 123,
     (30, 8, 1): 70, (30, 8, 2): 69, (30, 1, 1): 230, (30, 1, 2): 121, (30, 1, 3): 86, (30, 1, 4): 67, (30, 1, 5): 79, (30, 1, 6): 76, (30, 1, 8): 61,
     (30, 1, 12): 61, (30, 1, 16): 59,
     (120, 1, 1): 234, (120, 1, 2): 199, (120, 1, 3): 195, (120, 1, 4): 189, (120, 1, 5): 200, (120, 1, 6): 199, (120, 1, 8): 195, (120, 1, 12): 207,
     (120, 1, 16): 205, (120, 8, 1): 241, (120, 8, 2): 235, (480, 1, 1): 675}
# W half-step resident (L = 8192, D = 512): (units, g): us
W = {(60, 2): 215, (60, 3): 166, (60, 4): 118, (60, 5): 123, (60, 6): 128, (60, 8): 109, (60, 16): 117,
     (30, 2): 215, (30, 3): 164, (30, 4): 114, (30, 5): 114, (30, 6): 89, (30, 8): 65, (30, 16): 63,
     (120, 2): 219, (120, 3): 221, (120, 4): 198, (120, 5): 212, (120, 6): 214, (120, 8): 198, (120, 16): 220, (480, 4): 713}

def err(model, data, P):
    e = 0.0
    for key, v in data.items():
        e += ((model(key) - v) / v) ** 2
    return (e / len(data)) ** 0.5

if __name__ == "__main__":
    best = None
    for fix, c16, a, pw, launch in itertools.product((3, 6, 9, 12, 15), (0.66, 0.69, 0.72, 0.75, 0.78, 0.81, 0.84), (0.07, 0.15, 0.2, 0.25, 0.3), (1.1, 1.2, 1.3), (2, 5)):
        P = (fix, c16, a, pw, launch)
        e = err(lambda k: stream(k[0], 512, 8192, k[1], k[2], P), H, P)
        if best is None or e < best[0]: best = (e, P)
    print("streaming: rms rel. error %.3f with (fix, c16, alone, perwave, launch) =" % best[0], best[1])
    P = best[1]
    for k in sorted(H): print("  ", k, H[k], "model %.0f" % stream(k[0], 512, 8192, k[1], k[2], P))
    best = None
    for st0, st1, c16, pfix, launch in itertools.product((2, 4, 6, 8), (6, 9, 12, 15), (0.62, 0.66, 0.70, 0.74), (0, 1, 2, 3), (2, 5)):
        P = (st0, st1, c16, pfix, launch)
        e = err(lambda k: res(k[0], 8192, 512, k[1], P), W, P)
        if best is None or e < best[0]: best = (e, P)
    print("resident: rms rel. error %.3f with (stage0, stage1, c16, pair_fix, launch) =" % best[0], best[1])
    P = best[1]
    for k in sorted(W): print("  ", k, W[k], "model %.0f" % res(k[0], 8192, 512, k[1], P))
