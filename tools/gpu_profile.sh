#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh r02 > gpurun_out/profile_r02.log 2>&1
tail -5 gpurun_out/profile_r02.log
ls gpurun_out/profile_r02
