#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace: total per step, and by the duration of the kernel
that follows the gap (are the launch-latency-bound coarse levels waiting for the host or for the GPU's own dispatch?).
usage: gaps.py <kernel_trace.csv> <steps>"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
steps = float(sys.argv[2])
rows.sort()
busy = sum(e - s for s, e, _ in rows) / 1e6
span = (rows[-1][1] - rows[0][0]) / 1e6
buckets = {"<2us": [0, 0.0], "2-5us": [0, 0.0], "5-10us": [0, 0.0], "10-50us": [0, 0.0], ">50us": [0, 0.0]}
after_small = [0, 0.0]
for (s0, e0, _), (s1, e1, _) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if g <= 0:
        continue
    k = "<2us" if g < 2 else "2-5us" if g < 5 else "5-10us" if g < 10 else "10-50us" if g < 50 else ">50us"
    buckets[k][0] += 1
    buckets[k][1] += g
    if g < 50 and (e1 - s1) / 1e3 < 30:
        after_small[0] += 1
        after_small[1] += g
print(f"kernels {len(rows)}  busy {busy:.1f} ms  span {span:.1f} ms  ({steps:.0f} steps: busy {busy / steps:.2f} ms/step)")
for k, (n, t) in buckets.items():
    print(f"  gaps {k:>8s}: {n / steps:8.1f} per step  {t / 1e3 / steps:7.3f} ms per step")
print(f"  gaps < 50 us in front of kernels shorter than 30 us: {after_small[0] / steps:.1f} per step, {after_small[1] / 1e3 / steps:.3f} ms per step")
