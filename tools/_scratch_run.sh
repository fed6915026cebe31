#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4za; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
