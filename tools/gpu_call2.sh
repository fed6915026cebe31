#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/call2
mkdir -p $O
(timeout 200 tools/bin/lstm_seq_bench 32 1024 256 2 3; timeout 60 tools/bin/lstm_seq_bench 5 37 40 2 1; timeout 60 tools/bin/lstm_seq_bench 37 50 300 1 1; timeout 60 tools/bin/lstm_seq_bench 16 64 512 2 1; timeout 60 tools/bin/lstm_seq_bench 64 128 256 2 1) > $O/lstm_seq.log 2>&1
grep -E "LSTM layer|ms  =|protocol|fault|MISMATCH|exceed" $O/lstm_seq.log | head -120
timeout 200 tools/bin/gemm_stagger_bench 30 > $O/gemm_stagger.log 2>&1
cat $O/gemm_stagger.log
timeout 300 python tools/diag_grads.py > $O/diag_full.log 2>&1; cat $O/diag_full.log | grep -v amdgpu.ids
DB=4 DT=64 DMODES=philox timeout 200 python tools/diag_grads.py > $O/diag_small.log 2>&1; cat $O/diag_small.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 200 python tools/bench_rnn.py --gen lstm > $O/rnn_lstm.log 2>&1; tail -1 $O/rnn_lstm.log
