#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4x; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -m gpu -x -q -k "head or golden or cfg2 or philox or full_size" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs"
for r in 1 2; do
  GT_HEAD_VEC=1 timeout 120 $B > $O/vec$r.json 2> $O/vec$r.err
  GT_HEAD_VEC=0 timeout 120 $B > $O/sca$r.json 2> $O/sca$r.err
done
tail -3 $O/pytest.log
