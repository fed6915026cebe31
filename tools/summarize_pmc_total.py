#!/usr/bin/env python
"""HBM bytes per step of a tools/bench_rnn.py configuration from its two PMC passes (gpurun_out/profile_<tag>/pmc_<gen>_<dtype>_{FETCH,WRITE}_SIZE):
    summarize_pmc_total.py <tag> <gen> <dtype> <steps in the pass incl. warm-up> <ms/step of the un-profiled run> <algorithmic GFLOP per step>
Appends a roofline section to profiles/<tag>_<gen>_<dtype>_summary.md (FETCH_SIZE is doubled: gfx950 counts a wide coalesced read
at half its bytes, MI355X_MICROARCH.md "HBM"; both counters are in KiB)."""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, gen, dtype, steps, ms, gflop = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
tot = collections.defaultdict(float)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(ROOT, "gpurun_out", "profile_" + tag, "pmc_%s_%s_%s" % (gen, dtype, c), "k_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        v = float(r["Counter_Value"]) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
        tot[c] += v
        per[r["Kernel_Name"].replace("void ", "").replace("gt::", "").split("(")[0]][c] += v
fetch, write = tot["FETCH_SIZE"] / steps, tot["WRITE_SIZE"] / steps
L = ["\n## HBM traffic and roofline (PMC passes of the same command, per G+D step)\n",
     "* FETCH (x2 corrected) **%.2f GB**, WRITE **%.2f GB** per step -> %.2f GB / %.2f ms = **%.2f TB/s** = %.2f of the 8 TB/s HBM3E peak"
     % (fetch / 1e9, write / 1e9, (fetch + write) / 1e9, ms, (fetch + write) / ms / 1e9, (fetch + write) / ms / 1e9 / 8.0),
     "* algorithmic %.0f GFLOP per step (SURVEY 8(d) formula) / %.2f ms = **%.1f TFLOP/s** = %.4f of the 2.5 PFLOP/s dense bf16-MFMA peak"
     % (gflop, ms, gflop / ms, gflop / ms / 2500.0),
     "\n| kernel | FETCH GB/step (x2) | WRITE GB/step |", "|---|---:|---:|"]
for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:10]:
    L.append("| `%s` | %.3f | %.3f |" % (k, v["FETCH_SIZE"] / steps / 1e9, v["WRITE_SIZE"] / steps / 1e9))
out = os.path.join(ROOT, "profiles", "%s_%s_%s_summary.md" % (tag, gen, dtype))
open(out, "a").write("\n".join(L) + "\n")
print("\n".join(L))
