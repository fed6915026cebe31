#!/bin/bash
# A/B of one environment switch on the cfg2 step: tools/gpu_ab.sh VAR VALUE_A VALUE_B   (GPU box; run through gpurun)
O=gpurun_out/ab; mkdir -p $O
VAR=$1; shift
for v in "$@" "$@"; do
  env $VAR=$v timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_$v.json") if l.startswith("{")][-1]
r=d["roofline"]
print("$VAR=$v ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), [(x["kernel"][16:40], round(x["avg_us"],1), x["launches_per_step"]) for x in r["variants"]])
PY
done
