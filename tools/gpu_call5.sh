#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call5
mkdir -p $O
for cfg in "0 0" "100 1" "100 2" "100 3" "50 3" "200 3"; do
  set -- $cfg
  GT_GEMM_STAGGER_TICKS=$1 GT_GEMM_STAGGER_MODE=$2 timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_$1_$2.json')); print('ticks $1 mode $2:', round(d['ms_per_step'],4), 'ms  gemm family', round(d['roofline']['gemm_family']['achieved'],1), 'TF', [ (v['kernel'][16:-1], round(v['avg_us'],1)) for v in d['roofline']['variants']])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_lstm -o k -- python $GRAFT_REPO_ROOT/tools/bench_rnn.py --gen lstm --steps 3 > $GRAFT_REPO_ROOT/$O/prof_lstm.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $O/prof_lstm.log
f=$(find $O/prof_lstm -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-200
