mkdir -p gpurun_out/r4g; O=gpurun_out/r4g
b() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/bench_$n.json") if l.startswith("{")][-1]
    r=d["roofline"]
    print("$n ms/step %.4f frac %.3f family %.3f"%(d["ms_per_step"], r["frac"], r["gemm_family"]["frac"]))
except Exception as e: print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
PY
}
b pitched
b dense --dense-x
GT_OPT_FUSED=0 b pitched_nofuse
b pitched2
b b4 --batch 4
b b8 --batch 8
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
