mkdir -p gpurun_out/r4c; O=gpurun_out/r4c
./tools/bin/unaligned_probe > $O/probe.txt 2>&1; cat $O/probe.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
b() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/bench_$n.json") if l.startswith("{")][-1]
    r=d["roofline"]
    print("$n ms/step %.4f frac %.3f family %.3f"%(d["ms_per_step"], r["frac"], r["gemm_family"]["frac"]), [(x["kernel"][16:36], round(x["avg_us"],1), x["launches_per_step"]) for x in r["variants"]])
except Exception as e: print("$n failed", e)
PY
}
b mono GT_HIP_LIB=$PWD/tools/bin/libgantts_mono.so
b split_aligned GT_GEMM_UNALIGNED=0
b split_unaligned GT_GEMM_UNALIGNED=1
b mono2 GT_HIP_LIB=$PWD/tools/bin/libgantts_mono.so
b split_aligned2 GT_GEMM_UNALIGNED=0
b split_unaligned2 GT_GEMM_UNALIGNED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
