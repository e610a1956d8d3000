#!/bin/bash
# GPU box, round 4: the pair long-run body in one- and two-wave workgroups (a CU refills wave by wave instead of four waves at a time).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04y; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --sm 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --graph osc 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --samples 1024 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --streams 262144 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --streams 131072 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --streams 786432 0,0,0,0 2,64,128,256 2,64,64,256
$S --sm --graph df1 0,0,0,0 1,128,128,256 1,128,64,256
$S --sm --graph cascade2 0,0,0,0 1,128,128,256 1,128,64,256
$S --sm --streams 262144 --graph cascade2 0,0,0,0 1,128,128,256 1,128,64,256
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
