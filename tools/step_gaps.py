#!/usr/bin/env python
"""Idle time of the step stream from a rocprofv3 kernel trace: step_gaps.py <kernel_trace.csv> [first_kernel_substring]
Splits the trace into steps at every launch whose name contains the substring (default: the first kernel of apply_generator,
the split first layer has none -- the first NT product of the generator), skips the first third (warm-up) and prints, per step:
wall time first start -> last end, sum of kernel durations, idle = wall - union of the kernel intervals, launches."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "optim_step_kernel"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# a step ends with the generator's optimizer launch: the 2nd launch of `key` closes it
steps, cur, seen = [], [], 0
for e in ev:
    cur.append(e)
    if key in e[2]:
        seen += 1
        if seen == 2:
            steps.append(cur)
            cur, seen = [], 0
steps = steps[len(steps) // 3:]
tot = {"wall": 0.0, "busy": 0.0, "union": 0.0, "n": 0}
gaps = {}
for st in steps:
    wall = st[-1][1] - st[0][0]
    busy = sum(b - a for a, b, _ in st)
    union, end = 0, st[0][0]
    for i, (a, b, n) in enumerate(st):
        if a > end and i:
            gaps.setdefault((st[i - 1][2].split("(")[0][-48:], n.split("(")[0][-48:]), []).append(a - end)
        if b > end:
            union += b - max(a, end)
            end = b
    tot["wall"] += wall; tot["busy"] += busy; tot["union"] += union; tot["n"] += len(st)
k = float(len(steps))
print("%d steps: wall %.1f us/step, sum of kernels %.1f, union %.1f, idle %.1f, launches %.1f"
      % (len(steps), tot["wall"] / k / 1e3, tot["busy"] / k / 1e3, tot["union"] / k / 1e3, (tot["wall"] - tot["union"]) / k / 1e3, tot["n"] / k))
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  gap %6.2f us/step (%4.1f x %5.2f us)  %s -> %s" % (sum(v) / k / 1e3, len(v) / k, sum(v) / len(v) / 1e3, a, b))
