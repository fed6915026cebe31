#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call8
mkdir -p $O
rm -f $O/parity_report.txt
GT_PARITY_REPORT=$O/parity_report.txt timeout 300 python -m pytest tests -m gpu -q -k "bf16" > $O/pytest_bf16_report.log 2>&1
cat $O/parity_report.txt | grep bf16 | head -60; tail -5 $O/pytest_bf16_report.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|rror" $O/pytest.log | head
for dt in fp32 bf16; do
timeout 200 python tools/bench_rnn.py --gen lstm --dtype $dt > $O/rnn_lstm_$dt.log 2>&1; tail -1 $O/rnn_lstm_$dt.log
timeout 200 python tools/bench_rnn.py --gen sru --dtype $dt > $O/rnn_sru_$dt.log 2>&1; tail -1 $O/rnn_sru_$dt.log
timeout 200 python tools/bench_rnn.py --gen mlp --frames 512 --steps 20 --dtype $dt > $O/rnn_mlp_$dt.log 2>&1; tail -1 $O/rnn_mlp_$dt.log
done
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --force-dp > $O/bench_dp_engine.json 2> $O/bench_dp.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_dp_engine.json
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_plain.json 2> $O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_plain.json
