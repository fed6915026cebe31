#!/bin/bash
# round-2 GPU call 1: persistent LSTM A/B, full GPU test-suite in report mode, short benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/call1
mkdir -p $O
rm -f $O/parity_report.txt
(timeout 150 tools/bin/lstm_seq_bench 32 1024 256 2 3; timeout 60 tools/bin/lstm_seq_bench 5 37 40 2 1; timeout 60 tools/bin/lstm_seq_bench 37 50 300 1 1; timeout 60 tools/bin/lstm_seq_bench 16 64 512 2 1) > $O/lstm_seq.log 2>&1
tail -60 $O/lstm_seq.log
GT_PARITY_REPORT=$O/parity_report.txt timeout 900 python -m pytest tests -m gpu -q > $O/pytest_report.log 2>&1
tail -15 $O/pytest_report.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -30 $O/pytest.log
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
timeout 200 python tools/bench_rnn.py --gen lstm > $O/rnn_lstm.log 2>&1; tail -2 $O/rnn_lstm.log
GT_LSTM_STEPS=1 timeout 200 python tools/bench_rnn.py --gen lstm > $O/rnn_lstm_steps.log 2>&1; tail -2 $O/rnn_lstm_steps.log
