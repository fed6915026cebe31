#!/bin/bash
O=gpurun_out/call22; mkdir -p $O
timeout 300 python tools/bench_cfg5.py > $O/cfg5.log 2>&1; tail -3 $O/cfg5.log | cut -c1-400
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-dp > $O/bench_dp.json 2> $O/bench_dp.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_dp.json
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench.json
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --dp-python --force-dp > $O/bench_dpp.json 2> $O/bench_dpp.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_dpp.json
