#!/usr/bin/env bash
# Round 6, batch j (run on several boards):   gpurun --timeout 600 -- 'bash tools/experiments/exp_r06j.sh'   -> gpurun_out/r06j/
#  config 2 (65 536 x 4096) in interleaved bursts: the stage-packed wave against one and two I/O waves, plain rows and tiles of 8192, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1
for pass in 1 2; do
  timeout 300 python tools/sweep.py --graph cascade6 --streams 65536 --rounds 9 --reps 50 0,0 1,16,0,32768 1,16,0,33587200 >> $O/config2.txt 2>&1
  timeout 300 python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 9 --reps 50 0,0 1,16,0,32768 1,16,0,33587200 >> $O/config2.txt 2>&1
done
grep -v amdgpu.ids $O/config2.txt | cut -c1-170
