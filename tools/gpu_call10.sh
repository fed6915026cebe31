#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call10
mkdir -p $O
rm -f $O/parity_report.txt
GT_PARITY_REPORT=$O/parity_report.txt timeout 300 python -m pytest tests -m gpu -q -k "bf16" > $O/pytest_bf16_report.log 2>&1
grep -E "bf16 (y_hat|d_scal|g_scal)" $O/parity_report.txt; grep "bf16 update" $O/parity_report.txt | sort -k5 -g | tail -3; tail -3 $O/pytest_bf16_report.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|rror" $O/pytest.log | head
for i in 1 2; do
timeout 200 python tools/bench_rnn.py --gen lstm --dtype bf16 > $O/rnn_lstm_bf16.log 2>&1; tail -1 $O/rnn_lstm_bf16.log | cut -c1-100
timeout 200 python tools/bench_rnn.py --gen lstm --dtype fp32 > $O/rnn_lstm_fp32.log 2>&1; tail -1 $O/rnn_lstm_fp32.log | cut -c1-100
done
timeout 200 python tools/bench_rnn.py --gen mlp --frames 512 --steps 20 --dtype bf16 > $O/rnn_mlp_bf16.log 2>&1; tail -1 $O/rnn_mlp_bf16.log | cut -c1-100
timeout 200 python tools/bench_rnn.py --gen sru --dtype bf16 > $O/rnn_sru_bf16.log 2>&1; tail -1 $O/rnn_sru_bf16.log | cut -c1-100
