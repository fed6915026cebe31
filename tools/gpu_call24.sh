#!/bin/bash
O=gpurun_out/call24; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log | head -1
GT_GEMM_CHAIN=1 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -2 $O/pytest_chain.log | head -1
GT_GEMM_TILES=big timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_big.log 2>&1; tail -2 $O/pytest_big.log | head -1
timeout 200 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
