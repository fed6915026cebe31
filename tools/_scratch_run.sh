#!/bin/bash
# scratch driver for one gpurun call (not part of the product; overwritten per experiment)
set -u
O=gpurun_out/r4ipc; mkdir -p $O
export TMPDIR=/tmp
echo "HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-unset}" > $O/pytest.log
timeout 400 python -m pytest tests/test_gpu_comm2.py -m gpu -x -q -k "ipc" >> $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
