#!/bin/bash
# GPU box, round 4: config 2 (65 536 streams) on plain time-major frames: the stage-packed lone waves of a CU in lockstep / XCD-synchronised (the walk in
# step was only ever tried from 262 144 streams on), and 131 072 streams.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04s; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 11"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --streams 65536 0,0,0,0 1,16,256,$((L+8)) 1,16,256,$((LG+8)) 1,8,256,$((LG+8)) 1,4,256,$((LG+8)) 1,32,256,$((LG+8)) 1,16,256,33587200
$S --streams 131072 0,0,0,0 1,16,512,$((LG+8)) 1,8,512,$((LG+8)) 1,16,256,$((LG+8)) 2,8,256,$LG 2,4,256,$LG
$S --streams 65536 --tile 8192 0,0,0,0 1,16,256,$((L+8)) 1,16,256,33587200
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
