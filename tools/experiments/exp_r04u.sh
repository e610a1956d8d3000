#!/bin/bash
# GPU box, round 4: wide frames in lockstep -- ONE 1024-lane workgroup per CU (FZ_VF_MAX_WG(1): four laps at 1 M streams) against the two that fit its
# 43 registers (two laps): one lap of 256 workgroups did 0.80 at 262 144 streams, one lap of 512 (two per CU) 0.765 at 524 288.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04u; mkdir -p $O
MODE=${1:-run}
LG=8912896; LGP=8912928; W1=1048576
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph par4 0,0,0,0 1,2,1024,$((LG+W1)) 1,1,1024,$((LGP+W1)) 1,4,1024,$((LG+W1))
$S --graph par4 --streams 524288 0,0,0,0 1,2,1024,$((LG+W1)) 1,1,1024,$((LGP+W1)) 1,1,1024,$LGP
$S --graph par4 --streams 1000000 0,0,0,0 1,1,1024,$((LGP+W1)) 1,2,1024,$((LG+W1))
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
