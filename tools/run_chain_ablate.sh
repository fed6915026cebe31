#!/bin/bash
# builds and runs the panel-chain ablations on the GPU box (per-stage clock stamps of workgroup 0)
cd "$(dirname "$0")"
for flags in "" "-DCH_ABL_NOBAR" "-DCH_ABL_NOSTORE" "-DCH_ABL_NOLOAD" "-DCH_ABL_NOSTORE -DCH_ABL_NOLOAD -DCH_ABL_NOBAR"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCH_DEBUG_TIMING $flags chain_bench.hip -o /tmp/cb 2>/dev/null
  echo "== flags: [$flags]"
  timeout 60 /tmp/cb | grep -v "wg  255\|wg  511" | head -4
done
