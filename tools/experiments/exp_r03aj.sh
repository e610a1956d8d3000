#!/bin/bash
# round 3: one workgroup per CU for the few-stream kernels (FZ_VF_MAX_WG(1) = 1048576: LDS padding) -- does the dispatcher stack workgroups?
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03aj; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
M=1048576
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,32,64,$((34816+M)) 1,32,64,2048 1,32,64,$((2048+M)) 1,16,64,$((34816+M)) 2>&1 | grep -v amdgpu.ids > $O/config2q.txt
python tools/sweep.py --graph cascade6 --streams 8192 --tile 8192 --rounds 40 0,0 1,32,64,34816 1,32,64,$((34816+M)) 2>&1 | grep -v amdgpu.ids > $O/config2e.txt
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,$((33792+M)) 1,32,64,$((33792+M)) 1,32,64,33792 2>&1 | grep -v amdgpu.ids > $O/config2h.txt
tail -n +1 $O/config2*.txt
