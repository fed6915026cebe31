#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call7
mkdir -p $O
getms() { python - "$1" <<'PY'
import json,sys
txt=open(sys.argv[1]).read()
line=[l for l in txt.splitlines() if l.startswith("{")][-1]
d=json.loads(line); print(round(d["ms_per_step"],4), "ms", [round(v,5) for v in d["last_step_scalars"]["g"]])
PY
}
for i in 1 2; do
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_plain$i.json 2> $O/bench.err; echo -n "plain       : "; getms $O/bench_plain$i.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --force-dp > $O/bench_dp_engine$i.json 2> $O/bench_dp.err; echo -n "engine comm : "; getms $O/bench_dp_engine$i.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --force-dp --dp-python > $O/bench_dp_python$i.json 2> $O/bench_dp2.err; echo -n "python dp   : "; getms $O/bench_dp_python$i.json
done
GT_SB_HALF_ZERO=1 timeout 200 tools/bin/gemm_stagger_bench 20 > $O/gemm_stagger_halfzero.log 2>&1; grep -E "mixed" $O/gemm_stagger_halfzero.log
(timeout 100 tools/bin/lstm_seq_bench 32 1024 256 2 3) > $O/lstm.log 2>&1; grep -E "xcd-local bt8|MISMATCH" $O/lstm.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED" $O/pytest.log | head
timeout 200 python tools/bench_rnn.py --gen lstm > $O/rnn_lstm.log 2>&1; tail -1 $O/rnn_lstm.log
timeout 200 python tools/bench_rnn.py --gen sru > $O/rnn_sru.log 2>&1; tail -1 $O/rnn_sru.log
