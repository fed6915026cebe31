#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r04      -> gpurun_out/profile_r04/{stats,pmc_*}/... + cfg_<name>/ (tools/profile_configs.sh)
# --kernel-trace --stats in one run; PMC counters in their own runs (never combined with sys/hip tracing).
# tools/summarize_profile.py + tools/summarize_configs.py condense the result into profiles/<tag>_*.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# no clock spin-up KERNELS in the trace (they are launches of a product kernel and would be counted into its row); the clock is warmed by 350
# real steps instead (0.45 s), so that the traced averages are taken at the clock of the un-profiled line (round 6: 108.4 -> 98.8 us for the
# dominant kernel between a cold 25-step trace and a warm one)
BENCH="python $ROOT/bench.py --steps 50 --warmup 350 --no-cpu-baseline --no-other-configs --no-companion --spinup-ms 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/stats.log
BENCHP="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --spinup-ms 0"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o k -- $BENCHP > /dev/null 2> $OUT/pmc_fetch.log
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o k -- $BENCHP > /dev/null 2> $OUT/pmc_write.log
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -o k -- $BENCHP > /dev/null 2> $OUT/pmc_mfma.log
cd $ROOT
# the un-profiled lines: the default run (with other_configs and the CPU baseline), the driver's command, the clock spin-up A/B,
# the data-parallel schedule with one rank, dense x, the small-batch proxy of strong scaling
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --spinup-ms 0 --no-cpu-baseline --no-other-configs > $OUT/bench_driver_nospin.json 2> /dev/null
timeout 200 python bench.py --steps 50 --warmup 10 --force-dp --no-cpu-baseline > $OUT/bench_force_dp.json 2> $OUT/bench_force_dp.err
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_plain2.json 2> /dev/null      # (the plain line again, next to the data-parallel ones)
timeout 200 python bench.py --steps 50 --warmup 10 --force-dp --dp-ipc --no-cpu-baseline > $OUT/bench_force_dp_ipc.json 2> $OUT/bench_force_dp_ipc.err
timeout 200 python bench.py --steps 50 --warmup 10 --dense-x --no-cpu-baseline --no-other-configs > $OUT/bench_dense_x.json 2> /dev/null
# the data-parallel step's communication SCHEDULE (engine trace; tools/summarize_comm_schedule.py): one rank over RCCL, two ranks on this one
# device over the tests' RCCL double, and the same over the interprocess arenas
timeout 200 python bench.py --force-dp --comm-trace 20 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/sched_dp1_rccl.json 2> $OUT/sched_dp1_rccl.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --one-device --comm-trace 20 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/sched_dp2_fake.json 2> $OUT/sched_dp2_fake.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --one-device --dp-ipc --comm-trace 20 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/sched_dp2_ipc.json 2> $OUT/sched_dp2_ipc.err
for b in 16 8 4; do timeout 200 python bench.py --steps 50 --warmup 10 --batch $b --no-cpu-baseline --no-other-configs > $OUT/bench_b$b.json 2> /dev/null; done
bash $ROOT/tools/profile_configs.sh $TAG
cd $ROOT
for f in bench bench_driver bench_driver_nospin bench_force_dp bench_plain2 bench_force_dp_ipc bench_dense_x bench_b16 bench_b8 bench_b4; do python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$OUT/$f.json") if l.startswith("{")][-1]
    print("$f", round(d["ms_per_step"],4), round(d["roofline"]["frac"],3) if d.get("roofline") else None)
except Exception as e: print("$f failed", e)
PY
done
