#!/bin/bash
O=gpurun_out/call23; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "lstm or gru or rnn or bf16" > $O/pytest_lstm.log 2>&1; tail -4 $O/pytest_lstm.log
for m in rs gather; do
  if [ $m = gather ]; then export GT_LSTM_BWD_GATHER=1; else unset GT_LSTM_BWD_GATHER; fi
  echo "bwd $m"
  timeout 200 python tools/bench_rnn.py --gen lstm --dtype fp32 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen lstm --dtype bf16 2>&1 | tail -1 | cut -c1-90
done
