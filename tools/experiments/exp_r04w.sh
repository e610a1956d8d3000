#!/bin/bash
# GPU box, round 4: LDS rings -- half a wave per SIMD is all the rings leave room for (96 slots x 8 bytes x 128 lanes = 98 KiB: one two-wave workgroup per CU),
# so the rows in flight per wave are what hides the HBM latency: three chunk buffers (FZ_VF_PREFETCH3 = 32), four streams per lane.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04w; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph ldsring 0,0,0,0 2,32,128,32 1,32,256,32 2,16,128,32 4,16,64,0 4,32,64,0 4,16,64,32 2,32,64,32 1,32,128,32
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
