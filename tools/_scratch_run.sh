#!/bin/bash
# scratch A/B driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "riders or golden or train_loop or philox" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs"
for r in 1 2; do
  GT_LAUNCH_RIDERS=1 timeout 120 $B > $O/on$r.json 2> $O/on$r.err
  GT_LAUNCH_RIDERS=0 timeout 120 $B > $O/off$r.json 2> $O/off$r.err
done
for r in 1 2; do
  timeout 120 $B --steps 20 --warmup 5 > $O/drv400_$r.json 2> $O/drv400_$r.err
  timeout 120 $B --steps 20 --warmup 5 --spinup-ms 2000 > $O/drv2000_$r.json 2> $O/drv2000_$r.err
done
GT_LAUNCH_RIDERS=1 timeout 120 $B --batch 4 > $O/b4_on.json 2> $O/b4_on.err
GT_LAUNCH_RIDERS=0 timeout 120 $B --batch 4 > $O/b4_off.json 2> $O/b4_off.err
tail -3 $O/pytest.log
