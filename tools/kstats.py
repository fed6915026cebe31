#!/usr/bin/env python
"""Prints per-step kernel times from a rocprofv3 k_kernel_stats.csv: kstats.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    ps = float(r["TotalDurationNs"]) / steps / 1e3
    tot += ps
    if ps >= 2.0:
        print("%-70s n/step %5.1f avg %8.1f us  per-step %8.1f us" % (r["Name"].replace("gt::", "").replace("void ", "").split("(")[0][:70],
                                                                  float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ps))
print("total GPU-busy per step: %.1f us" % tot)
