#!/bin/bash
O=gpurun_out/call28; mkdir -p $O
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 64 --frames 777 > $O/bench_odd.json 2> $O/bench_odd.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_odd.json | head -1; tail -2 $O/bench_odd.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 5 --frames 100 > $O/bench_small.json 2> $O/bench_small.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_small.json | head -1; tail -2 $O/bench_small.err
timeout 300 python tools/bench_rnn.py --gen sru --batch 16 --frames 2048 --dtype fp32 2>&1 | tail -1 | cut -c1-100
timeout 300 python tools/bench_rnn.py --gen sru --batch 16 --frames 2048 --dtype bf16 2>&1 | tail -1 | cut -c1-100
timeout 300 python tools/bench_rnn.py --gen lstm --batch 64 --frames 512 --dtype fp32 2>&1 | tail -1 | cut -c1-100
bash tools/profile_round.sh r02 > gpurun_out/profile_r02.log 2>&1
tail -c 300 gpurun_out/profile_r02/bench.json
