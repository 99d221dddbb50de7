#!/usr/bin/env python3
"""Round 4: would SELL-C-sigma cut the padding of the blocked sparse form's sliced ELL?
BASELINE configs[3] matrix (100000 x 4096, 0.5 % fill, uniformly random positions).  A slice = 64 lane elements (one per lane),
a run per (slice, granule of 1024 rows of the gathered factor), run length = the slice's longest lane element in that granule.
A lane element keeps its lane for the whole half-step (its factor row and numerators live in that lane's registers), so a
permutation of the lane elements applies to ALL granules at once.  Prints slots per non-zero for: the order as given; lane
elements sorted by total length inside windows of sigma (SELL-C-sigma); the per-granule lower bound if every granule could
be sorted on its own (it cannot); granules merged in pairs / fours (fewer, longer runs)."""
import numpy as np

n, m, fill = 100000, 4096, 0.005
rng = np.random.default_rng(3)
nnz = int(n * m * fill)
rows, cols = rng.integers(0, n, nnz), rng.integers(0, m, nnz)
key = np.unique(rows.astype(np.int64) * m + cols)
rows, cols = key // m, key % m
nz = len(key)


def counts(lane_el, gathered, L, D, G=1024):
    ngb = (D + G - 1) // G
    c = np.zeros((L, ngb), np.int32)
    np.add.at(c, (lane_el, gathered // G), 1)
    return c


def slots(c):  # c: [lane elements in slice order][granules]
    L = c.shape[0]
    pad = (-L) % 64
    if pad:
        c = np.vstack([c, np.zeros((pad, c.shape[1]), c.dtype)])
    return int(c.reshape(-1, 64, c.shape[1]).max(axis=1).sum()) * 64


for name, le, ga, L, D in (("W half-step (lane elements = rows, 4 granules of columns)", rows, cols, n, m),
                            ("H half-step (lane elements = columns, 98 granules of rows)", cols, rows, m, n)):
    c = counts(le, ga, L, D)
    print(name)
    print(f"  as given:                               {slots(c) / nz:.2f} slots per non-zero")
    for sigma in (256, 1024, 8192, L):
        order = np.concatenate([s + np.argsort(-c[s:s + sigma].sum(axis=1), kind="stable") for s in range(0, L, sigma)])
        print(f"  sorted by total length, sigma = {sigma:6d}:  {slots(c[order]) / nz:.2f}")
    best = sum(slots(np.sort(c[:, b:b + 1], axis=0)[::-1]) for b in range(c.shape[1]))
    print(f"  every granule sorted on its own (bound): {best / nz:.2f}   <- needs a lane element to change lanes between granules")
    for g in (2, 4):
        cg = np.add.reduceat(c, np.arange(0, c.shape[1], g), axis=1)
        print(f"  runs over {g} granules ({g * 1024} rows staged at once): {slots(cg) / nz:.2f}   <- LDS holds {g} granules only for k <= {16 if g == 2 else 4}")
