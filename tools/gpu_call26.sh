#!/bin/bash
O=gpurun_out/call26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 4 2 1; do
GT_MLPG_FPL=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$a -o k -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $O/p$a.log
python - <<PY
import csv
for r in csv.DictReader(open('$O/p$a/k_kernel_stats.csv')):
    if 'mlpg_forward' in r['Name'] or 'mlpg_backward' in r['Name']: print($a, r['Name'][:34], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
GT_MLPG_FPL=$a timeout 300 python -m pytest tests -m gpu -x -q -k "mlpg or acoustic_mlp" 2>&1 | tail -1
done
