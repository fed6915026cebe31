#!/usr/bin/env python
"""profiles/<tag>_<name>_summary.md from a `rocprofv3 --kernel-trace --stats` run of tools/bench_rnn.py:
    summarize_rnn_profile.py <tag> <name> <stats.csv> <steps traced> <bench_rnn log> [<command line>]"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, path, steps, log = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
cmd = sys.argv[6] if len(sys.argv) > 6 else ""
rows = list(csv.DictReader(open(path)))
line = [l for l in open(log).read().splitlines() if "ms/step" in l]
L = ["# %s -- rocprofv3 kernel trace of `%s` (1x MI355X)\n" % (tag, cmd or ("tools/bench_rnn.py " + name)),
     "Un-profiled run of the same command: `%s`\n" % (line[-1].split(", scalars")[0] if line else "n/a"),
     "Per G+D step (%d steps in the trace, the first one includes one-time work):\n" % steps,
     "| kernel | launches/step | avg us | ms/step | % |", "|---|---:|---:|---:|---:|"]
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    tot += ms
    if float(r["Percentage"]) < 0.2:
        continue
    nm = r["Name"].replace("void ", "").replace("gt::", "").split("(")[0]
    L.append("| `%s` | %.1f | %.1f | %.3f | %.1f |" % (nm, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms, float(r["Percentage"])))
L.append("\nGPU-busy time per step (sum of all kernels): **%.2f ms**.\n" % tot)
out = os.path.join(ROOT, "profiles", "%s_%s_summary.md" % (tag, name))
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
