#!/bin/bash
O=gpurun_out/call20; mkdir -p $O
for t in 128 64; do
  export GT_BF16_TILES=$t
  timeout 200 python tools/bench_rnn.py --gen mlp --frames 512 --steps 20 --dtype bf16 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen lstm --dtype bf16 2>&1 | tail -1 | cut -c1-90
  timeout 200 python tools/bench_rnn.py --gen sru --dtype bf16 2>&1 | tail -1 | cut -c1-90
done
unset GT_BF16_TILES
GT_BF16_TILES=64 timeout 300 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -2
