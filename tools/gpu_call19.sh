#!/bin/bash
O=gpurun_out/call19; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_$i.json") if l.startswith("{")][-1]
r=d["roofline"]
print("ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), r["kernel"], round(r["frac"],3), [(v["kernel"][16:], round(v["avg_us"],1), v["launches_per_step"]) for v in r["variants"]])
PY
done
