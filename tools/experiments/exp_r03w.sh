#!/bin/bash
# round 3: what do the barriers of the wave-split rounds cost?  (-DFZ_DBG_NOBARRIER: wrong results, time only)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03w; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
for mode in base nobar; do
  if [ $mode = nobar ]; then export FLOWZ_HIP_EXTRA_OPTS="-DFZ_DBG_NOBARRIER"; fi
  python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 2>&1 | grep -v amdgpu > $O/config2_$mode.txt
  python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,1024 1,16,128,34816 1,32,128,33792 2>&1 | grep -v amdgpu > $O/config2h_$mode.txt
  python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,16,64,34816 1,32,64,2048 2>&1 | grep -v amdgpu > $O/config2q_$mode.txt
done
tail -n +1 $O/*.txt
