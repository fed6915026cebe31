#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4sq; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-other-configs --trace-steps"
for r in 1 2 3; do
  timeout 120 $B --steps 20 --warmup 5 > $O/drv$r.json 2> $O/drv$r.err
done
timeout 120 $B > $O/s50.json 2> $O/s50.err
timeout 120 $B --batch 4 > $O/b4.json 2> $O/b4.err
timeout 120 $B --steps 3 --warmup 1 > $O/tiny.json 2> $O/tiny.err
grep "host ms" $O/drv*.err
