#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01      -> gpurun_out/profile_r01/{stats,pmc_*}/...
# --kernel-trace --stats in one run; PMC counters in their own runs (never combined with sys/hip tracing).
set -u
TAG=${1:-r03}
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
for v in "lstm fp32 1024 3" "lstm bf16 1024 3" "sru fp32 1024 3" "sru bf16 1024 3" "mlp bf16 512 20"; do
  set -- $v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rnn_$1_$2 -o k -- python $ROOT/tools/bench_rnn.py --gen $1 --dtype $2 --frames $3 --steps $4 > /dev/null 2> $OUT/rnn_$1_$2.err
  timeout 200 python $ROOT/tools/bench_rnn.py --gen $1 --dtype $2 --frames $3 --steps $(( $4 * 2 )) > $OUT/rnn_$1_$2.log 2>&1
done
# HBM bytes of the bf16-storage configurations (BASELINE.json configs[2]): FETCH_SIZE / WRITE_SIZE in their own passes
for v in "lstm bf16 1024 3" "mlp bf16 512 6"; do
  set -- $v
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$1_$2_$c -o k -- python $ROOT/tools/bench_rnn.py --gen $1 --dtype $2 --frames $3 --steps $4 > /dev/null 2> $OUT/pmc_$1_$2_$c.err
  done
done
# cfg4 / cfg5 shaped lines
timeout 200 python $ROOT/tools/bench_rnn.py --gen sru --dtype fp32 --batch 16 --frames 2048 --steps 3 > $OUT/cfg4_fp32.log 2>&1
timeout 200 python $ROOT/tools/bench_rnn.py --gen sru --dtype bf16 --batch 16 --frames 2048 --steps 3 > $OUT/cfg4_bf16.log 2>&1
timeout 200 python $ROOT/tools/bench_cfg5.py > $OUT/cfg5.log 2>&1
timeout 200 python $ROOT/bench.py --steps 50 --warmup 10 --force-dp --no-cpu-baseline > $OUT/bench_force_dp.json 2> $OUT/bench_force_dp.err
cd $ROOT
ls -R $OUT | head -40
tail -c 600 $OUT/bench.json
