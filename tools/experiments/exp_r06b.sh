#!/usr/bin/env bash
# Round 6, second GPU batch:   gpurun --timeout 3000 -- 'bash tools/experiments/exp_r06b.sh'   -> gpurun_out/r06b/
#  the whole GPU suite + the default bench line on the tree after the subtraction (tools/gpu_round.sh: manifest recorded, JIT-built objects harvested),
#  then the placement experiment again with a warm-up and interleaved rounds (tools/experiments/exp_r06_placement.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_round.sh r06b
O=$R/gpurun_out/r06b
export FLOWZ_HIP_NO_PLAN_CACHE=1
for g in ldsring osc blocks64; do timeout 900 python tools/experiments/exp_r06_placement.py $g > $O/placement_$g.txt 2>&1; cat $O/placement_$g.txt | cut -c1-420; done
