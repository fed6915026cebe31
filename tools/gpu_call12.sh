#!/bin/bash
mkdir -p gpurun_out/call12
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/call12/pytest.log 2>&1; tail -3 gpurun_out/call12/pytest.log
for i in 1 2; do
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/call12/bench_$i.json 2> gpurun_out/call12/bench_$i.err
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/call12/prof -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/call12/prof.log 2>&1
python tools/kstats.py $(find gpurun_out/call12/prof -name "*kernel_stats.csv" | head -1) 25 > gpurun_out/call12/kstats.txt 2>&1
python - <<'PY'
import json
for n in ["1","2"]:
    try:
        d=[json.loads(l) for l in open("gpurun_out/call12/bench_%s.json"%n) if l.startswith("{")][-1]
        r=d["roofline"]
        print(n, "ms/step %.4f"%d["ms_per_step"], "family frac %.3f ms %.3f"%(r["gemm_family"]["frac"], r["gemm_family"]["ms_per_step"]))
    except Exception as e: print(n, "ERR", e)
PY
head -40 gpurun_out/call12/kstats.txt
