mkdir -p gpurun_out/r4e; O=gpurun_out/r4e
b() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/bench_$n.json") if l.startswith("{")][-1]
    r=d["roofline"]
    print("$n ms/step %.4f frac %.3f family %.3f"%(d["ms_per_step"], r["frac"], r["gemm_family"]["frac"]), [(x["kernel"][16:52], round(x["avg_us"],1), x["launches_per_step"]) for x in r["variants"]])
except Exception as e: print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
PY
}
b split1 GT_D_SPLIT=1
b split0 GT_D_SPLIT=0
b split1b GT_D_SPLIT=1
b split0b GT_D_SPLIT=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
