#!/bin/bash
# rocprofv3 evidence for the BASELINE.json configurations besides the headline (bench.py: other_configs), on the GPU box:
#   tools/profile_configs.sh r04 [names...]   -> gpurun_out/profile_<tag>/cfg_<name>/{stats,pmc_fetch,pmc_write}/ + <name>.json
# One --kernel-trace --stats run, then FETCH_SIZE and WRITE_SIZE each in their own --pmc run (never combined with tracing
# domains other than the kernel trace), then the un-profiled line.  tools/summarize_configs.py condenses them into profiles/.
set -u
TAG=${1:-r04}; shift || true
NAMES=${@:-cfg1 cfg3_bf16 cfg3_fp32 cfg4_bf16 cfg5 cfg2_bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in $NAMES; do
  D=$OUT/cfg_$n; mkdir -p $D
  CMD="python $ROOT/tools/bench_configs.py $n --steps 3 --warmup 1"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o k -- $CMD > /dev/null 2> $D/stats.log
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o k -- $CMD > /dev/null 2> $D/pmc_fetch.log
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o k -- $CMD > /dev/null 2> $D/pmc_write.log
  timeout 300 python $ROOT/tools/bench_configs.py $n --steps 10 --warmup 2 > $D/line.json 2> $D/line.err
  tail -c 400 $D/line.json; echo
done
