#!/usr/bin/env python
"""Average duration of the kernels whose name contains one of the substrings, last two thirds of a rocprofv3 kernel trace:
kernel_avgs.py <kernel_trace.csv> <substring> [...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for key in sys.argv[2:]:
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
    v = v[len(v) // 3:]
    if v:
        print("%-40s n %4d avg %7.2f us  min %7.2f" % (key, len(v), sum(v) / len(v), min(v)))
