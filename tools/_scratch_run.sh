#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs"
for r in 1 2; do
  timeout 200 $B --force-dp > $O/dp$r.json 2> $O/dp$r.err
  timeout 120 $B > $O/plain$r.json 2> $O/plain$r.err
done
GT_LAUNCH_RIDERS=0 timeout 200 $B --force-dp > $O/dp_norid.json 2> $O/dp_norid.err
timeout 200 $B --force-dp --batch 4 > $O/dp_b4.json 2> $O/dp_b4.err
tail -3 $O/pytest.log
