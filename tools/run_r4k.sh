mkdir -p gpurun_out/r4k; O=gpurun_out/r4k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_gpu_comm2.py -m gpu -q -k "sru or SRU or cfg4" > $O/pytest_sru.log 2>&1; echo "pytest sru rc=$?"; tail -3 $O/pytest_sru.log
for cw in 64 32; do
  GT_SRU_CW=$cw timeout 300 python tools/bench_configs.py cfg4_bf16 --steps 10 --warmup 2 > $O/cfg4_cw$cw.json 2>/dev/null
  python -c "
import json; d=json.loads([l for l in open('$O/cfg4_cw$cw.json') if l.startswith('{')][-1]); print('cfg4_bf16 CW=$cw', round(d['cfg4_bf16']['ms_per_step'],3))"
  GT_SRU_CW=$cw timeout 300 python tools/bench_rnn.py --gen sru --dtype bf16 --steps 10 2>/dev/null | cut -c1-90
  GT_SRU_CW=$cw timeout 300 python tools/bench_rnn.py --gen sru --dtype fp32 --steps 5 2>/dev/null | cut -c1-90
done
b() { n=$1; shift
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json; d=[json.loads(l) for l in open('$O/bench_$n.json') if l.startswith('{')][-1]; print('$n', round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"; }
b side
GT_SIDE_OVERLAP=0 b noside
b side2
GT_SIDE_OVERLAP=0 b noside2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
