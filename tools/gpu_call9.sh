#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call9
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|rror" $O/pytest.log | head
for dt in bf16; do
timeout 200 python tools/bench_rnn.py --gen lstm --dtype $dt > $O/rnn_lstm_$dt.log 2>&1; tail -1 $O/rnn_lstm_$dt.log | cut -c1-120
timeout 200 python tools/bench_rnn.py --gen sru --dtype $dt > $O/rnn_sru_$dt.log 2>&1; tail -1 $O/rnn_sru_$dt.log | cut -c1-120
timeout 200 python tools/bench_rnn.py --gen mlp --frames 512 --steps 20 --dtype $dt > $O/rnn_mlp_$dt.log 2>&1; tail -1 $O/rnn_mlp_$dt.log | cut -c1-120
done
timeout 200 python tools/bench_rnn.py --gen sru --batch 16 --frames 2048 > $O/rnn_sru_cfg4.log 2>&1; tail -1 $O/rnn_sru_cfg4.log | cut -c1-120
timeout 200 python tools/bench_rnn.py --gen sru --batch 16 --frames 2048 --dtype bf16 > $O/rnn_sru_cfg4_bf16.log 2>&1; tail -1 $O/rnn_sru_cfg4_bf16.log | cut -c1-120
