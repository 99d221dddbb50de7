#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and reports how much the step kernels overlap in time."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
step = [r for r in rows if "step_kernel" in r[2]]
print("kernels", len(rows), "step kernels", len(step), "queues", sorted(set(r[3] for r in step)))
t0, t1 = step[0][0], max(r[1] for r in step)
busy = sum(r[1] - r[0] for r in step)
print(f"span {1e-6 * (t1 - t0):.3f} ms, sum of step-kernel durations {1e-6 * busy:.3f} ms, avg concurrency {busy / (t1 - t0):.2f}")
by = collections.defaultdict(list)
for r in step:
    by[r[2].split("step_kernel")[1][:12]].append(r[1] - r[0])
for k, v in sorted(by.items()):
    print(f"  {k:14s} n={len(v):4d} avg={1e-3 * sum(v) / len(v):8.1f} us")
# timeline of the first 40 step kernels
for r in step[:40]:
    print(f"  {1e-3 * (r[0] - t0):9.1f} -> {1e-3 * (r[1] - t0):9.1f} us  q={r[3]} {r[2].split('step_kernel')[1][:12]}")
