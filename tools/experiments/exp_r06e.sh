#!/usr/bin/env bash
# Round 6, batch e:   gpurun --timeout 900 -- 'bash tools/experiments/exp_r06e.sh'   -> gpurun_out/r06e/
#  the LDS rings in lockstep with THREE chunk buffers (batch d: +- 1 % over six allocations against +- 1.9 % with two) on another board, two passes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_CACHE=/tmp/fz_kc_r06e R06D_THREE_BUFFERS=1
for pass in 1 2; do timeout 600 python tools/experiments/exp_r06d.py >> $O/ldsring_three_buffers.txt 2>&1; done
grep -v amdgpu.ids $O/ldsring_three_buffers.txt | cut -c1-600
