#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call6
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "communicator or dp2 or rccl" > $O/pytest_dp.log 2>&1
grep -E "passed|failed|FAILED|Error|error" $O/pytest_dp.log | head -20
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_plain.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_plain.json')); print('plain        :', round(d['ms_per_step'],4), 'ms')"
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --force-dp > $O/bench_dp_engine.json 2> $O/bench_dp.err; tail -3 $O/bench_dp.err; python -c "
import json; d=json.load(open('$O/bench_dp_engine.json')); print('engine comm  :', round(d['ms_per_step'],4), 'ms', d['last_step_scalars'])"
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --force-dp --dp-python > $O/bench_dp_python.json 2> $O/bench_dp2.err; python -c "
import json; d=json.load(open('$O/bench_dp_python.json')); print('python dp    :', round(d['ms_per_step'],4), 'ms')"
GT_GEMM_STAGGER_TICKS=3200 GT_GEMM_STAGGER_MODE=2 timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_st3200.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_st3200.json')); print('stagger 32us :', round(d['ms_per_step'],4), 'ms')"
timeout 200 tools/bin/gemm_stagger_bench 20 > $O/gemm_stagger.log 2>&1; grep -E "mixed" $O/gemm_stagger.log
(timeout 100 tools/bin/lstm_seq_bench_nosent 32 1024 256 2 3) > $O/lstm_nosent.log 2>&1; grep -E "xcd-local bt8|MISMATCH" $O/lstm_nosent.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED" $O/pytest.log | head
