#!/usr/bin/env bash
# Round 6, batch h (run on several boards):   gpurun --timeout 600 -- 'bash tools/experiments/exp_r06h.sh'   -> gpurun_out/r06h/
#  stream tiles of 8192, 1 M streams: the library's free-running default (two streams per lane, two workgroups per CU) against the lockstep geometries on tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1
L=524288; G=8388608; LG=$((L+G))
for g in cascade6 osc par4f; do
  timeout 300 python tools/sweep.py --graph $g --tile 8192 --rounds 7 --reps 3 0,0 2,2,1024,$LG 4,1,1024,$((LG+32)) 1,4,1024,$LG 2,16,256,0 >> $O/tiled_$(hostname).txt 2>&1
done
grep -v amdgpu.ids $O/tiled_$(hostname).txt | cut -c1-200
