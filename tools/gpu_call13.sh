#!/bin/bash
mkdir -p gpurun_out/call13
timeout 300 python -m pytest tests -m gpu -x -q -k "philox or acoustic_mlp" > gpurun_out/call13/pytest_first.log 2>&1; tail -5 gpurun_out/call13/pytest_first.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/call13/pytest.log 2>&1; tail -5 gpurun_out/call13/pytest.log
for v in 0 1 0 1; do
GT_GEMM_CHAIN=$v timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call13/bench_$v.json 2> gpurun_out/call13/bench_$v.err
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/call13/bench_$v.json") if l.startswith("{")][-1]
r=d["roofline"]
print("chain=$v ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), [(v["kernel"][16:], round(v["avg_us"],1), v["launches_per_step"]) for v in r["variants"]])
PY
done
