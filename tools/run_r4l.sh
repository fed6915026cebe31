mkdir -p gpurun_out/r4l; O=gpurun_out/r4l
b() { n=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json; d=[json.loads(l) for l in open('$O/bench_$n.json') if l.startswith('{')][-1]; r=d['roofline']; print('$n', round(d['ms_per_step'],4), round(r['frac'],3), [(x['kernel'][16:30], round(x['avg_us'],1)) for x in r['variants'] if 'pair' in x['kernel']])" || tail -3 $O/bench_$n.err; }
b drv1 --steps 20 --warmup 5
b drv_nospin --steps 20 --warmup 5 --spinup-ms 0
b drv2 --steps 20 --warmup 5
b s50 --steps 50 --warmup 10
GT_TN_SPLIT_WGS=1152 b s50_tn1152 --steps 50 --warmup 10
GT_TN_SPLIT_WGS=768 b s50_tn768 --steps 50 --warmup 10
GT_TN_SPLIT_WGS=2048 b s50_tn2048 --steps 50 --warmup 10
b b16 --steps 50 --warmup 10 --batch 16
b b16b --steps 200 --warmup 20 --batch 16
b b24 --steps 50 --warmup 10 --batch 24
b b12 --steps 50 --warmup 10 --batch 12
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or split or pitched or philox" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
