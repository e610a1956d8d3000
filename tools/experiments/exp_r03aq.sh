#!/bin/bash
# round 3: more rows in flight for the synchronised time-major walk -- 512-lane workgroups (256 registers per lane), two laps
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03aq; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
L=524288; G=8388608; PF=32
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 4,2,512,$((L+G)) 4,4,512,$((L+G)) 4,2,512,$((L+G+PF)) 4,4,512,$((L+G+PF)) 4,8,512,$((L+G)) 2,4,512,$((L+G)) 2,8,512,$((L+G)) 2,4,1024,$((L+G)) 4,2,1024,$((L+G)) 2,8,512,$((L+G+PF)) 2>&1 | grep -v amdgpu.ids > $O/timemajor_cascade6.txt
python tools/sweep.py --graph df1 --streams 1048576 --tile 0 --rounds 7 0,0 4,1,1024,$((L+G+PF)) 4,4,512,$((L+G)) 4,8,512,$((L+G)) 4,4,512,$((L+G+PF)) 2,8,512,$((L+G)) 2>&1 | grep -v amdgpu.ids > $O/timemajor_df1.txt
tail -n +1 $O/*.txt
