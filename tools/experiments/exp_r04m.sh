#!/bin/bash
# GPU box, round 4: (1) the SLP vectoriser (FZ_VF_SLP, flag 4) on the 4-wire sum: four isomorphic biquads side by side, which it pairs into packed operations
# (ISA: 295 v_pk_add + 290 v_pk_mul where the scalar body has 551 + 580) -- stream-major hold body, plain time-major, tiles; (2) stream-major, 65 536
# streams: one-wave workgroups against four-wave workgroups once more (r04k and r04l disagree); (3) hold body, other shapes.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04m; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --sm --graph par4 0,0,0,0 1,32,256,4 1,32,128,0 1,32,128,4 1,32,64,4
$S --graph par4 0,0,0,0 1,32,256,4 1,16,256,4 2,16,256,0
$S --graph par4 --tile 4096 0,0,0,0 1,32,256,4 1,16,256,4
$S --sm --graph par4 --streams 262144 0,0,0,0 1,32,256,4
$S --sm --streams 65536 1,128,64,264 1,128,256,264 1,128,128,264
$S --sm --streams 65536 1,128,256,264 1,128,64,264
$S --sm --graph par4f 0,0,0,0 1,128,256,260 
$S --graph par4f 0,0,0,0 4,1,1024,$((8912928+4)) 2,2,1024,$((8912896+4))
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
