#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call4
mkdir -p $O
(timeout 200 tools/bin/lstm_seq_bench 32 1024 256 2 3; timeout 60 tools/bin/lstm_seq_bench 5 37 40 2 1; timeout 60 tools/bin/lstm_seq_bench 37 50 300 1 1; timeout 60 tools/bin/lstm_seq_bench 16 64 512 2 1) > $O/lstm_seq.log 2>&1
grep -E "LSTM layer|ms  =|protocol|MISMATCH|exceed|HIP error" $O/lstm_seq.log | head -150
timeout 200 tools/bin/gemm_stagger_bench 30 > $O/gemm_stagger.log 2>&1
grep -E "ideal|mode" $O/gemm_stagger.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest.log | head
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.load(open('$O/bench.json')); print('bench stagger on :', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['gemm_family'])"
GT_GEMM_STAGGER_TICKS=0 timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_nostagger.json 2> $O/bench2.err; python -c "
import json,sys; d=json.load(open('$O/bench_nostagger.json')); print('bench stagger off:', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['gemm_family'])"
timeout 200 python tools/bench_rnn.py --gen lstm > $O/rnn_lstm.log 2>&1; tail -1 $O/rnn_lstm.log
