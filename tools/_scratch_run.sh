#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4y; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-other-configs"
for r in 1 2; do
  GT_RES_HOSTMAP=1 timeout 120 $B > $O/map$r.json 2> $O/map$r.err
  GT_RES_HOSTMAP=0 timeout 120 $B > $O/copy$r.json 2> $O/copy$r.err
done
cd /tmp
GT_RES_HOSTMAP=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o copy -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --spinup-ms 0 --no-roofline > /dev/null 2>&1
GT_POLL_RESULTS=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o poll -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --spinup-ms 0 --no-roofline > /dev/null 2>&1
