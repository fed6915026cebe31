#!/bin/bash
mkdir -p gpurun_out/call15
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/call15/pytest.log 2>&1; tail -5 gpurun_out/call15/pytest.log
for v in 0 1; do
GT_GEMM_CHAIN=$v timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call15/bench_$v.json 2> gpurun_out/call15/bench_$v.err
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/call15/bench_$v.json") if l.startswith("{")][-1]
r=d["roofline"]
print("chain=$v ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), r["kernel"], round(r["frac"],3))
PY
done
