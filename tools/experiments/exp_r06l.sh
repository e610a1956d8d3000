#!/usr/bin/env bash
# Round 6, batch l (run on two boards):   gpurun --timeout 600 -- 'bash tools/experiments/exp_r06l.sh'   -> gpurun_out/r06l/
#  32 768 and 16 384 streams: the wave-split defaults (one I/O wave) against the same splits with a loader AND a storer wave, and the new one-compute-wave arrangement
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1
S="timeout 300 python tools/sweep.py --graph cascade6 --rounds 9 --reps 60"
IO=32768; IO2=33587200; W2=1024; W3=2048
$S --streams 32768 0,0 1,32,0,$((W2+IO2)) 1,16,0,$((W2+IO2)) 1,32,0,$W2 1,16,0,$IO2 1,32,0,$((W3+IO)) 1,32,0,$((W3+IO2)) >> $O/few_streams.txt 2>&1
$S --streams 32768 --tile 8192 0,0 1,32,0,$((W2+IO2)) 1,16,0,$IO2 >> $O/few_streams.txt 2>&1
$S --streams 16384 0,0 1,32,0,$((W3+IO2)) 1,16,0,$((W3+IO2)) 1,32,0,$((W2+IO2)) 1,32,0,$((W2+IO)) >> $O/few_streams.txt 2>&1
grep -v amdgpu.ids $O/few_streams.txt | cut -c1-170
