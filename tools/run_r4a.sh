mkdir -p gpurun_out/r4a; O=gpurun_out/r4a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --spinup-ms 0 > $O/bench_driver_nospin.json 2> $O/bench_driver_nospin.err
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_50.json 2> $O/bench_50.err
tail -3 $O/pytest.log; cat $O/rc.txt
for f in bench_driver bench_driver_nospin bench_50; do python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$f.json") if l.startswith("{")][-1]
    print("$f", d["ms_per_step"], d["roofline"]["frac"], {k:(round(v.get("ms_per_step",-1),3) if "ms_per_step" in v else v) for k,v in d.get("other_configs",{}).items()})
except Exception as e: print("$f failed", e)
PY
done
tail -5 $O/bench_driver.err
