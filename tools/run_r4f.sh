mkdir -p gpurun_out/r4f; O=gpurun_out/r4f
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
b split1b GT_D_SPLIT=1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline > /dev/null 2> $GRAFT_REPO_ROOT/$O/stats.log
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=0
for r in rows[:40]:
    n=r["Name"][:90]; c=int(r["Calls"]); t=float(r["TotalDurationNs"])/25/1e3
    tot+=t
    print("%-90s %5.1f/step %7.1f us avg %7.1f us/step"%(n, c/25, float(r["AverageNs"])/1e3, t))
print("total/step", tot)
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_gpu_comm2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
