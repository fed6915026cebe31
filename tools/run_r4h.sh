mkdir -p gpurun_out/r4h; O=gpurun_out/r4h
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
b fused
GT_OPT_FUSED=0 b nofuse
b fused2
GT_OPT_FUSED=0 b nofuse2
GT_OPT_FUSED=0 b b4_nofuse --batch 4
b b4_fused --batch 4
timeout 600 python -m pytest tests/test_gpu_at_size.py -m gpu -x -q -k "cfg2_cold" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
GT_PARITY_REPORT=$PWD/$O/cold_report.txt timeout 600 python -m pytest tests/test_gpu_at_size.py -m gpu -x -q -k "cfg2_cold" > /dev/null 2>&1; cat $O/cold_report.txt | head -60
