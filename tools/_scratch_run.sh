#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4rep; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_report.txt
GT_PARITY_REPORT=$O/parity_report.txt timeout 1500 python -m pytest tests/test_gpu_at_size.py -m gpu -q > $O/pytest_at_size.log 2>&1; echo "rc=$?" >> $O/pytest_at_size.log
GT_PARITY_REPORT=$O/parity_report_philox.txt timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "philox_on or full_size or cfg2" > $O/pytest_philox.log 2>&1; echo "rc=$?" >> $O/pytest_philox.log
tail -15 $O/pytest_at_size.log; tail -5 $O/pytest_philox.log
