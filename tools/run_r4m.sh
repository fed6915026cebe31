mkdir -p gpurun_out/r4m; O=gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm2.py -m gpu -q -x -k "philox or golden or split" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
b() { n=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json; d=[json.loads(l) for l in open('$O/bench_$n.json') if l.startswith('{')][-1]; r=d['roofline']; print('$n', round(d['ms_per_step'],4), round(r['frac'],3), [(x['kernel'][16:40], round(x['avg_us'],1), x['launches_per_step']) for x in r['variants']])" || tail -3 $O/bench_$n.err; }
b seg --steps 50 --warmup 10
GT_SPLIT_FUSED=0 b noseg --steps 50 --warmup 10
b seg2 --steps 50 --warmup 10
GT_SPLIT_FUSED=0 b noseg2 --steps 50 --warmup 10
b b4 --steps 50 --warmup 10 --batch 4
