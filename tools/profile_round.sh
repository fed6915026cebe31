#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01      -> gpurun_out/profile_r01/{stats,pmc_*}/...
# --kernel-trace --stats in one run; PMC counters in their own runs (never combined with sys/hip tracing).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/stats.log
BENCHP="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o k -- $BENCHP > /dev/null 2> $OUT/pmc_fetch.log
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o k -- $BENCHP > /dev/null 2> $OUT/pmc_write.log
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -o k -- $BENCHP > /dev/null 2> $OUT/pmc_mfma.log
cd $ROOT
timeout 200 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
# recurrent generators (cfg3 BiLSTM fp32 / bf16 products, hparams-default SRU): kernel trace + the un-profiled line
cd /tmp
for v in "lstm fp32" "lstm bf16" "sru fp32"; do
  set -- $v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rnn_$1_$2 -o k -- python $ROOT/tools/bench_rnn.py --gen $1 --dtype $2 --steps 3 > /dev/null 2> $OUT/rnn_$1_$2.err
  timeout 200 python $ROOT/tools/bench_rnn.py --gen $1 --dtype $2 > $OUT/rnn_$1_$2.log 2>&1
done
cd $ROOT
ls -R $OUT | head -40
tail -c 600 $OUT/bench.json
