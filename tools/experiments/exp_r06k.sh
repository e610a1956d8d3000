#!/usr/bin/env bash
# Round 6, batch k (run on two boards):   gpurun --timeout 600 -- 'bash tools/experiments/exp_r06k.sh'   -> gpurun_out/r06k/
#  the new default between 32 768 and 65 536 streams (one stage-packed compute wave + two I/O waves) against the lone stage-packed wave on the shapes batch j did not cover:
#  other stream counts, the oscillator chain (31 per-stream coefficients, a scalar prefix), cascades of 4 and 8 stages
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1
S="timeout 300 python tools/sweep.py --rounds 9 --reps 50"
for ns in 49152 40960 36864; do $S --graph cascade6 --streams $ns 0,0 1,16,0,8 1,16,0,32768 >> $O/io2_default.txt 2>&1; done
$S --graph osc --streams 65536 0,0 1,16,0,8 1,16,0,32768 >> $O/io2_default.txt 2>&1
$S --graph osc --streams 65536 --tile 8192 0,0 1,16,0,8 >> $O/io2_default.txt 2>&1
$S --graph cascade4 --streams 65536 0,0 1,16,0,8 >> $O/io2_default.txt 2>&1
for g in cascade8 cascade10 cascade12 cascade2 df1; do $S --graph $g --streams 65536 0,0 1,16,0,8 1,16,0,33587200 >> $O/io2_default.txt 2>&1; done
$S --graph cascade6g --streams 65536 0,0 1,16,0,8 >> $O/io2_default.txt 2>&1
grep -v amdgpu.ids $O/io2_default.txt | cut -c1-170
