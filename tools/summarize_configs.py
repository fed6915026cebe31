#!/usr/bin/env python
"""profiles/<tag>_<cfg>_summary.md + profiles/<tag>_other_configs.json from gpurun_out/profile_<tag>/cfg_<name>/ (tools/profile_configs.sh):
per configuration the kernel trace (time per step and kernel), the HBM bytes per step and kernel from the FETCH_SIZE / WRITE_SIZE
passes (FETCH x 1024 x 2: gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md "HBM"; WRITE x 1024), the
un-profiled line of the same command, and both roofline fractions.  bench.py reads the JSON for `other_configs[*].roofline.hbm`."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
STEPS = 4.0          # bench_configs.py --steps 3 --warmup 1 under the profiler


def short(n):
    return n.replace("void ", "").replace("gt::", "").split("(")[0]


def counters(path, name):
    per = collections.defaultdict(float)
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return per
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name:
            per[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0 * (2.0 if name == "FETCH_SIZE" else 1.0)
    return per


summary = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "profile_" + tag, "cfg_*"))):
    name = os.path.basename(d)[4:]
    st = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    rows = list(csv.DictReader(open(st[0])))
    fetch, write = counters(os.path.join(d, "pmc_fetch"), "FETCH_SIZE"), counters(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    try:
        line = json.loads([l for l in open(os.path.join(d, "line.json")) if l.startswith("{")][-1])[name]
    except Exception:
        line = None
    hbm = (sum(fetch.values()) + sum(write.values())) / STEPS
    L = ["# %s -- %s: rocprofv3 evidence for `python tools/bench_configs.py %s` (1x MI355X)\n" % (tag, name, name)]
    if line:
        L.append("%s\n" % line["config"])
        L.append("* un-profiled run (10 steps after 2): **%.3f ms/step**, %.2f M frames/s, dtype %s" % (line["ms_per_step"], line["frames_per_s"] / 1e6, line["dtype"]))
        r = line["roofline"]
        L.append("* step-level MFMA roofline: %.0f GFLOP algorithmic (SURVEY 8(d)) / step -> %.1f TFLOP/s = **%.4f** of the %.0f TFLOP/s dense %s peak"
                 % (line["step_algorithmic_gflop"], r["achieved"], r["frac"], r["peak"], line["dtype"]))
        if hbm > 0:
            bw = hbm / (line["ms_per_step"] * 1e-3) / 1e12
            L.append("* HBM (PMC passes below): **%.2f GB/step** -> %.2f TB/s = **%.3f** of the 8 TB/s HBM3E peak" % (hbm / 1e9, bw, bw / 8.0))
            summary[name] = {"hbm_bytes_per_step": hbm, "ms_per_step_at_profile_time": line["ms_per_step"],
                             "hbm_frac": bw / 8.0, "mfma_frac": r["frac"]}
    L.append("\n## kernel trace (`--kernel-trace --stats`; %d steps in the trace, the first one includes one-time work) and HBM bytes per kernel\n" % STEPS)
    L.append("| kernel | launches/step | avg us | ms/step | % | FETCH GB/step (x2) | WRITE GB/step | GB/s while it runs |")
    L.append("|---|---:|---:|---:|---:|---:|---:|---:|")
    tot = 0.0
    for r in rows:
        ms = float(r["TotalDurationNs"]) / STEPS / 1e6
        tot += ms
        if float(r["Percentage"]) < 0.3:
            continue
        k = short(r["Name"])
        f, w = fetch.get(k, 0.0) / STEPS, write.get(k, 0.0) / STEPS
        L.append("| `%s` | %.1f | %.1f | %.3f | %.1f | %.3f | %.3f | %.0f |" % (k, float(r["Calls"]) / STEPS, float(r["AverageNs"]) / 1e3, ms,
                                                                           float(r["Percentage"]), f / 1e9, w / 1e9, (f + w) / max(ms, 1e-9) / 1e6))
    L.append("\nGPU-busy time per step (sum of all kernels in the trace): **%.2f ms**.\n" % tot)
    open(os.path.join(ROOT, "profiles", "%s_%s_summary.md" % (tag, name)), "w").write("\n".join(L) + "\n")
    print("\n".join(L[:12]))
json.dump(summary, open(os.path.join(ROOT, "profiles", "%s_other_configs.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
