#!/bin/bash
for t in big default; do
  if [ $t = big ]; then export GT_GEMM_TILES=big; else unset GT_GEMM_TILES; fi
  echo "tiles $t"
  timeout 200 python tools/bench_rnn.py --gen lstm --dtype fp32 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen sru --dtype fp32 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen sru --dtype bf16 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen lstm --dtype bf16 2>&1 | tail -1 | cut -c1-90
done
