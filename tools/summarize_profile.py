#!/usr/bin/env python
"""Condenses gpurun_out/profile_<tag>/ (written by tools/profile_round.sh on the GPU box) into the
tracked evidence under profiles/: kernel-trace stats, per-kernel PMC aggregates (HBM traffic with the
gfx950 FETCH_SIZE correction, MFMA busy fraction) and the bench.py JSON line of the same build."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("void ", "").replace("gt::", "").split("(")[0]


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return a, {k: len(v) for k, v in n.items()}


shutil.copy(os.path.join(src, "stats", "k_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)

stats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "k_kernel_stats.csv")))
         if "at::native" not in r["Name"] and not r["Name"].startswith("__amd_rocclr")]      # torch's own fills / copies of the set-up are not the step
_tot_ns = sum(float(r["TotalDurationNs"]) for r in stats) or 1.0
for r in stats:
    r["Percentage"] = "%.4f" % (100.0 * float(r["TotalDurationNs"]) / _tot_ns)
# steps in the trace = optimizer launches / 2 (one per network and G+D step; warm-up steps included -- since round 6 the traced command warms
# the clock with 350 REAL steps instead of the spin-up kernels, so that the trace holds nothing but the step and runs at the un-profiled clock)
_opt = [float(r["Calls"]) for r in stats if "optim_step_kernel" in r["Name"]]
steps = _opt[0] / 2.0 if _opt else 25.0
fetch, nf = agg(os.path.join(src, "pmc_fetch", "k_counter_collection.csv"))
write, nw = agg(os.path.join(src, "pmc_write", "k_counter_collection.csv"))
mfma, nm = agg(os.path.join(src, "pmc_mfma", "k_counter_collection.csv"))

L = []
L.append("# %s -- rocprofv3 evidence for `python bench.py` (cfg2, B=32, T=512, fp32, 1x MI355X)\n" % tag)
L.append("Collected by `tools/profile_round.sh %s` (one `--kernel-trace --stats` run; FETCH_SIZE, WRITE_SIZE and the "
         "SQ/GRBM counters each in their own `--pmc` run), condensed by `tools/summarize_profile.py`.\n" % tag)
L.append("## bench.py line of the same build (un-profiled run)\n")
L.append("* value: **%.3f M frames/s**, %.3f ms/step, step-level MFMA fraction %.3f (138.0 GFLOP algorithmic / step)"
         % (bench["value"] / 1e6, bench["ms_per_step"], bench["step_mfma_frac"]))
r = bench["roofline"]
L.append("* dominant kernel `%s`: %.1f TFLOP/s = %.3f of the %.1f TFLOP/s f32-MFMA peak, %.1f us per launch (HIP events)"
         % (r["kernel"], r["achieved"], r["frac"], r["peak"], r["avg_launch_us"]))
_dom = [float(x["AverageNs"]) / 1e3 for x in stats if "gemm_pair_kernel" in x["Name"]]
if _dom:      # the same kernel by the trace below (its own start-to-end time; the event bracket of the line above includes the dispatch)
    _fl = r["achieved"] * r["avg_launch_us"]      # TFLOP/s x us = MFLOP per launch
    L.append("* the same kernel by the rocprofv3 trace below: %.1f us per launch = %.1f TFLOP/s = **%.3f** of the peak (%d steps in the trace, "
             "350 of them warm-up: the clock of the un-profiled line)" % (_dom[0], _fl / _dom[0], _fl / _dom[0] / r["peak"], int(steps)))
L.append("* GEMM family: %.1f TFLOP/s (%.3f), %.3f ms of the step" % (r["gemm_family"]["achieved"], r["gemm_family"]["frac"],
                                                                    r["gemm_family"]["ms_per_step"]))
if "cpu_baseline" in bench:
    c = bench["cpu_baseline"]
    L.append("* cpu_baseline (%s, %d threads): %.0f frames/s -- %s" % (c["kind"], c["cores"], c["value"], c["sample"]))
L.append("\n## kernel trace (`--kernel-trace --stats`), per G+D step\n")
L.append("| kernel | launches/step | avg us | us/step | % |")
L.append("|---|---:|---:|---:|---:|")
tot = 0.0
for s in stats:
    per_step = float(s["TotalDurationNs"]) / steps / 1e3
    tot += per_step
    if float(s["Percentage"]) < 0.15:
        continue
    L.append("| `%s` | %.1f | %.1f | %.1f | %.1f |" % (short(s["Name"]), float(s["Calls"]) / steps,
                                                      float(s["AverageNs"]) / 1e3, per_step, float(s["Percentage"])))
L.append("\nGPU-busy time per step (sum of kernels): **%.0f us**.\n" % tot)
L.append("## PMC aggregates per launch (averages over the launches of each kernel)\n")
L.append("HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950 counts a wide coalesced read at half its bytes, "
         "MI355X_MICROARCH.md \"HBM\") + WRITE_SIZE x 1024.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
         "(GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs).\n")
L.append("| kernel | launches | FETCH MB (x2 corrected) | WRITE MB | HBM GB/s (bytes / avg launch time of the trace) | MFMA busy | MFMA instr / launch |")
L.append("|---|---:|---:|---:|---:|---:|---:|")
avg_us = {short(x["Name"]): float(x["AverageNs"]) / 1e3 for x in stats}
for k in sorted(mfma, key=lambda k: -mfma[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) - fetch.get(k, {}).get("FETCH_SIZE", 0)):
    f = fetch.get(k, {}).get("FETCH_SIZE", 0) / max(1, nf.get(k, 1)) * 1024 * 2 / 1e6
    w = write.get(k, {}).get("WRITE_SIZE", 0) / max(1, nw.get(k, 1)) * 1024 / 1e6
    m = mfma[k]
    n = max(1, nm[k])
    gui = m.get("GRBM_GUI_ACTIVE", 0) / n / 8.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n / (gui * 1024) if gui else 0.0
    if f + w < 0.5 and busy == 0:
        continue
    gbs = (f + w) * 1e6 / (avg_us[k] * 1e-6) / 1e9 if avg_us.get(k) else 0.0
    L.append("| `%s` | %d | %.1f | %.1f | %.0f | %.2f | %.0f |" % (k, n, f, w, gbs, busy, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n / 64))
# machine-readable PMC aggregates per GEMM variant (kind, BN) -- bench.py reports them as roofline.traffic
import re
pmc = {}
for k in mfma:
    mm = re.match(r"gemm_f32_kernel<(\d), (\d+), (\d+), (\w+), (\w+), (\d), (\d+), (-?\d+)", k)
    if mm:
        kind, am = int(mm.group(1)), int(mm.group(8))
        slot = {(0, 0): 6, (0, 1): 7, (0, 2): 9, (0, 4): 13, (1, 0): 10, (1, 1): 11}.get((kind, am))      # include/gantts_hip.h: GT_PROFILE_SLOTS layout
        key = "slot%d" % slot if slot is not None else "%s,%s" % (mm.group(1), mm.group(3))
    elif k.startswith("gemm_pair_kernel"):
        key = "pair"
    elif k.startswith("gemm_tn_pair_kernel"):
        key = "slot12"
    else:
        continue
    d = pmc.setdefault(key, {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    nl = max(1, nm[k])
    d["launches"] += nl
    d["fetch_bytes"] += fetch.get(k, {}).get("FETCH_SIZE", 0) / max(1, nf.get(k, 1)) * nl * 1024 * 2
    d["write_bytes"] += write.get(k, {}).get("WRITE_SIZE", 0) / max(1, nw.get(k, 1)) * nl * 1024
for d in pmc.values():
    d["hbm_bytes_per_launch"] = (d["fetch_bytes"] + d["write_bytes"]) / d["launches"]
    d["fetch_bytes"] /= d["launches"]
    d["write_bytes"] /= d["launches"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh %s); "
                     "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, averaged per launch" % tag,
           "gemm_variants": pmc}, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
