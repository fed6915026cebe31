#!/bin/bash
# builds and runs the GEMM ablations on the GPU box (NT 128x128, M=16384, N=512; K=512 and K=2048; full grid and half grid)
set -e
cd "$(dirname "$0")"
for flags in "" "-DGT_ABLATE_NO_LDSWRITE" "-DGT_ABLATE_NO_GLOBAL" "-DGT_ABLATE_NO_GLOBAL -DGT_ABLATE_NO_LDSREAD" "-DGT_ABLATE_NO_GLOBAL -DGT_ABLATE_NO_EPILOGUE" $EXTRA_ABLATIONS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVEC=true $flags gemm_ablate.hip -o /tmp/ablate 2>/dev/null
  echo "== flags: [$flags]"
  /tmp/ablate 512
  /tmp/ablate 2048
done
