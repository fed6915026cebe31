#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call3
mkdir -p $O
timeout 300 tools/bin/gemm_stagger_bench 30 > $O/gemm_stagger.log 2>&1
cat $O/gemm_stagger.log
rm -f $O/parity_report.txt
GT_PARITY_REPORT=$O/parity_report.txt timeout 600 python -m pytest tests -m gpu -q -k "philox or oracle_only" > $O/pytest_report.log 2>&1
grep -E "rel-rms|sru" $O/parity_report.txt | head -150
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
