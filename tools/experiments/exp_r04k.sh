#!/bin/bash
# GPU box, round 4: (1) stream-major buffers at 65 536 streams -- the PAIR long-run body on half the SIMDs (512 lone waves of 128 streams, 28
# instructions per step) against the stage-packed one-stream body on all of them (1024 lone waves of 64 streams, 30.4 + per step): a lone wave's
# time per step is what counts, not how many SIMDs are busy; (2) LDS rings with deeper chunks; (3) rows off the grid: load policies next to
# the nt store policy.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04k; mkdir -p $O
MODE=${1:-run}
LGP=8912928; NT=$((7<<16))
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --sm --streams 65536 0,0,0,0 2,64,256,256 2,64,128,256 2,64,64,256 1,128,64,264 1,128,128,264
$S --sm --streams 32768 0,0,0,0 2,64,64,256 2,64,128,256 1,128,64,264
$S --sm --streams 131072 0,0,0,0 2,64,64,256 2,64,128,256 2,64,256,256
$S --sm --streams 98304 0,0,0,0 2,64,64,256 2,64,128,256 2,64,256,256
$S --graph ldsring 0,0,0,0 2,32,128,0 2,64,128,0 1,64,256,0 1,32,256,0 1,32,128,0 2,32,64,0
$S --streams 1000001 4,1,1024,$LGP 4,1,1024,$((LGP+(1<<12))) 4,1,1024,$((LGP+(2<<12))) 4,1,1024,$((LGP+(5<<12)))
$S --streams 1048577 0,0,0,0
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
