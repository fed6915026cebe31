#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4w; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-other-configs --trace-steps"
timeout 120 $B --steps 20 --warmup 5 > $O/drv1.json 2> $O/drv1.err
timeout 120 $B --steps 20 --warmup 5 --no-roofline > $O/drv_noroof.json 2> $O/drv_noroof.err
timeout 120 $B --steps 20 --warmup 30 > $O/drv_w30.json 2> $O/drv_w30.err
timeout 120 $B --steps 50 --warmup 10 > $O/s50.json 2> $O/s50.err
timeout 120 $B --steps 20 --warmup 5 > $O/drv2.json 2> $O/drv2.err
grep "host ms" $O/*.err
