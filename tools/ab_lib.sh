#!/bin/bash
# A/B of two BUILDS of the library on the cfg2 step: tools/ab_lib.sh <other .so> [tag]   (GPU box; run through gpurun)
# bench.py twice per build, alternating; the other build is selected through GT_HIP_LIB (gantts_amd/_lib.py).
R=$(readlink -f $1); T=${2:-other}; O=gpurun_out/ab_$T; mkdir -p $O
for rep in 1 2; do for v in base $T; do
  if [ $v = base ]; then unset GT_HIP_LIB; else export GT_HIP_LIB=$R; fi
  timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_${v}_$rep.json") if l.startswith("{")][-1]
r=d["roofline"]
print("$v ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), [(x["kernel"][16:44], round(x["avg_us"],1), x["launches_per_step"]) for x in r["variants"]])
PY
done; done
