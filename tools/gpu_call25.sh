#!/bin/bash
O=gpurun_out/call25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 3; do
GT_MLPG_ABLATE=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$a -o k -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $O/p$a.log
echo "ablate $a: $(grep mlpg_forward $O/p$a/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120)"
done
