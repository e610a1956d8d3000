#!/bin/bash
# round 3: cache policies of the frame loads / stores for the time-major lockstep kernel (flags bits 12..14 loads, 16..18 stores:
# 1 = none, 2 = sc0, 3 = sc1, 4 = sc0 sc1, 5 = sc0 nt, 6 = sc1 nt, 7 = nt)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03l; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
B=$((524288+32))
V="4,1,1024,$B"
for l in 1 3 7; do V="$V 4,1,1024,$((B+(l<<12)))"; done
for s in 1 2 3 5 6 7; do V="$V 4,1,1024,$((B+(s<<16)))"; done
V="$V 4,1,1024,$((B+(1<<12)+(1<<16))) 4,1,1024,$((B+(3<<12)+(3<<16)))"
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 $V > $O/tm_cache_policies.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,0,32768 1,16,256,2048 1,24,256,8 > $O/config2_m1.txt 2>&1
echo done
