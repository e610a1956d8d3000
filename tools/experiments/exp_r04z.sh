#!/bin/bash
# GPU box, round 4: do the free-running frame kernels gain from smaller workgroups too (refill wave by wave)?  tiles and plain time-major rows.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z; mkdir -p $O
MODE=${1:-run}
W2=2097152
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --tile 8192 0,0,0,0 2,16,128,$W2 2,16,64,$W2 2,16,64,0 2,16,128,0
$S --graph par4 --tile 4096 0,0,0,0 1,32,128,0 1,32,64,0
$S --graph osc --tile 8192 0,0,0,0 2,16,128,$W2 2,16,64,$W2
$S --graph farring 0,0,0,0 1,8,128,0 1,8,64,0
$S --streams 65536 --tile 8192 0,0,0,0 1,16,64,8 1,16,128,8
$S --graph c32onepole 0,0,0,0 2,16,256,0 2,16,128,0 2,16,64,0
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
