#!/bin/bash
# GPU box, round 4: the 4-wire sum (config 3) on PLAIN TIME-MAJOR frames with the chip walking the rows in step (FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC, laps as
# launches) -- round 3 measured 0.75-0.77 for a persistent four-lap launch against 0.69-0.73 free-running and left it because tiles do 0.80.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04n; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896; LGP=8912928
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph par4 0,0,0,0 1,1,1024,$LGP 1,2,1024,$LG 1,4,1024,$LG 2,1,1024,$LGP 2,2,1024,$LG 1,4,512,$LG 1,2,512,$LG 1,8,256,$L 1,8,1024,$L
$S --graph par4 --streams 262144 0,0,0,0 1,1,1024,$LGP 1,2,1024,$LG 1,4,1024,$LG
$S --graph par4 --streams 1000000 0,0,0,0 1,2,1024,$LG
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
