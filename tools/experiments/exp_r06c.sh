#!/usr/bin/env bash
# Round 6, third GPU batch:   gpurun --timeout 3000 -- 'bash tools/experiments/exp_r06c.sh'   -> gpurun_out/r06c/
#  the whole GPU suite (comparison operators, LDS rings in lockstep, the 8-rank rehearsal) + the default bench line (tools/gpu_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_round.sh r06c
