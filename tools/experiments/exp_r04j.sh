#!/bin/bash
# GPU box, round 4: rows off the 16-byte grid once more -- is it the STORES?  A wave's footprint that does not start on a 32-byte sector
# leaves the sectors at its ends half-written; write-through stores (nt | sc1) send each half to HBM by itself (with ECC: a
# read-modify-write).  Plain write-back stores (flags + 1 << 16) let L2 merge the halves of neighbouring waves.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04j; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896; LGP=8912928; PL=$((1<<16)); NT=$((7<<16)); SC=$((3<<16))
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --streams 1000001 4,1,1024,$LGP 4,1,1024,$((LGP+PL)) 4,1,1024,$((LGP+NT)) 4,1,1024,$((LGP+SC))
$S --streams 1000002 4,1,1024,$LGP 4,1,1024,$((LGP+PL)) 2,2,1024,$LG 2,2,1024,$((LG+PL))
$S --streams 1000000 4,1,1024,$LGP 4,1,1024,$((LGP+PL))
$S --streams 1048576 4,1,1024,$LGP 4,1,1024,$((LGP+PL))
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
