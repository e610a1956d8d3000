#!/bin/bash
# GPU box, round 4: stream-major buffers at 262 144 streams and at 1024-row blocks: the one-stream body with 64-sample phases (two waves per SIMD) against the pair body.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04x; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --sm --streams 262144 0,0,0,0 1,64,256,264 1,128,256,264 1,64,128,264 2,64,128,256 2,64,64,256
$S --sm --samples 1024 0,0,0,0 1,64,256,264 1,128,256,264 2,64,128,256
$S --sm --streams 524288 0,0,0,0 1,64,256,264 2,64,128,256
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
