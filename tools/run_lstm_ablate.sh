#!/bin/bash
cd "$(dirname "$0")"
for flags in "" "-DGT_ABLATE_LSTM_NO_STAGE" "-DGT_ABLATE_LSTM_NO_MMA" "-DGT_ABLATE_LSTM_NO_GATES" "-DGT_ABLATE_LSTM_NO_STAGE -DGT_ABLATE_LSTM_NO_MMA -DGT_ABLATE_LSTM_NO_GATES"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags lstm_ablate.hip -o /tmp/lstm_ablate 2>/dev/null
  echo "== [$flags]"; /tmp/lstm_ablate | tail -1
done
