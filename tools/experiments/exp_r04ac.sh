#!/bin/bash
# (NOTE: the kernel / generator knob this script drives was an experiment and has been taken out again -- profiles/NOTES.md, "What the lone waves wait for"; kept as the record of what was run)
# GPU box, round 4: forwarded short ring reads against reads in place (FLOWZ_HIP_NO_RING_FORWARD=1), the same box, alternating.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04ac; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
for rep in 1 2; do
echo "## forwarded"
$S --graph ldsring 0,0,0,0 1,32,256,0 2,16,128,0 1,32,128,0
echo "## in place (FLOWZ_HIP_NO_RING_FORWARD=1)"
FLOWZ_HIP_NO_RING_FORWARD=1 $S --graph ldsring 0,0,0,0 1,32,256,0 2,16,128,0 1,32,128,0
done
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
