#!/bin/bash
O=gpurun_out/call16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
python tools/kstats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 25 > $O/kstats.txt 2>&1
head -45 $O/kstats.txt
timeout 200 python tools/bench_rnn.py --gen lstm --dtype fp32 > $O/rnn_lstm_fp32.log 2>&1; tail -1 $O/rnn_lstm_fp32.log | cut -c1-100
timeout 200 python tools/bench_rnn.py --gen lstm --dtype bf16 > $O/rnn_lstm_bf16.log 2>&1; tail -1 $O/rnn_lstm_bf16.log | cut -c1-100
timeout 200 python tools/bench_rnn.py --gen sru --dtype fp32 > $O/rnn_sru_fp32.log 2>&1; tail -1 $O/rnn_sru_fp32.log | cut -c1-100
