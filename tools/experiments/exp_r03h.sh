#!/bin/bash
# round 3: do the CUs of a launch drift apart on time-major frames?  the lockstep kernel on blocks of 128 ... 4096 samples
# (a short block gives the workgroups no time to drift), and the old default next to it
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03h; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
for T in 128 256 512 1024 2048 4096; do
  python tools/sweep.py --graph cascade6 --streams 1048576 --samples $T --tile 0 --rounds 9 0,0 2,16,256,0 > $O/tm_T$T.txt 2>&1
  python tools/sweep.py --graph cascade6 --streams 1048576 --samples $T --tile 8192 --rounds 9 0,0 > $O/tiled_T$T.txt 2>&1
done
echo done
