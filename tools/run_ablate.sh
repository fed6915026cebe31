#!/bin/bash
# builds and runs the GEMM ablations on the GPU box
set -e
cd "$(dirname "$0")"
for flags in "-DVEC=true" "-DVEC=false" "-DVEC=true -DGT_ABLATE_NO_EPILOGUE" "-DVEC=true -DGT_ABLATE_NO_GLOBAL"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags gemm_ablate.hip -o /tmp/ablate 2>/dev/null
  echo "== flags: [$flags]"
  /tmp/ablate 512
  /tmp/ablate 4096
done
