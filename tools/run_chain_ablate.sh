#!/bin/bash
# builds and runs the panel-chain ablations on the GPU box
cd "$(dirname "$0")"
for flags in "" "-DCH_ABL_NOBAR -DCH_ABL_NOLOAD -DCH_ABL_NOSTORE" "-DCH_ABL_NOOUT"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags chain_bench.hip -o /tmp/cb 2>/dev/null
  echo "== flags: [$flags]"
  timeout 60 /tmp/cb
done
