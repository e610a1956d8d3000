#!/bin/bash
# GPU box, round 4: pair body against the stage-packed one-stream body, both in one-wave workgroups, around the threshold of the pair rule (2^17 .. 2^19 streams).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04af; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 11"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
for n in 131072 262144 393216 524288; do $S --sm --streams $n 2,64,64,256 1,128,64,264; done
$S --sm --graph osc --streams 262144 2,64,64,256 1,128,64,264
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps_$(date +%s).txt 2>&1
grep -v amdgpu.ids $O/sweeps_*.txt | tail -20
