#!/bin/bash
# tile heuristic A/B: full GPU suite on the new default, then bench with big tiles vs 64x64 tiles
mkdir -p gpurun_out/call11
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/call11/pytest.log 2>&1; tail -3 gpurun_out/call11/pytest.log
GT_GEMM_TILES=big timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call11/bench_big.json 2> gpurun_out/call11/bench_big.err
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call11/bench_64.json 2> gpurun_out/call11/bench_64.err
GT_GEMM_TILES=big timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call11/bench_big2.json 2>/dev/null
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call11/bench_642.json 2>/dev/null
python - <<'PY'
import json
for n in ["big","64","big2","642"]:
    try:
        d=[json.loads(l) for l in open("gpurun_out/call11/bench_%s.json"%n) if l.startswith("{")][-1]
        r=d["roofline"]
        print(n, "ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]), [(v["kernel"][-22:], round(v["avg_us"],1), v["launches_per_step"]) for v in r["variants"]])
    except Exception as e: print(n, "ERR", e)
PY
