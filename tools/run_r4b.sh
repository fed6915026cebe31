mkdir -p gpurun_out/r4b; O=gpurun_out/r4b
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -5 $O/pytest.log; cat $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver.json 2> $O/bench_driver.err
python - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_driver.json") if l.startswith("{")][-1]
print("bench", d["ms_per_step"], d["roofline"]["frac"])
PY
