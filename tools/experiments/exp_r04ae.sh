#!/bin/bash
# (NOTE: the kernel / generator knob this script drives was an experiment and has been taken out again -- profiles/NOTES.md, "What the lone waves wait for"; kept as the record of what was run)
# GPU box, round 4: -DFZ_DBG_PRIME_VMCNT (dummy stores in the preheader of the chunk loop: the waitcnt pass's merged state at the loop header becomes exact)
# against the plain kernels, alternating processes on one box: kernels of one or two waves per SIMD.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04ae; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
for rep in 1 2 3; do
for opt in "" "-DFZ_DBG_PRIME_VMCNT"; do
echo "## EXTRA_OPTS=[$opt]"
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --graph ldsring 0,0,0,0 1,32,256,0
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --graph par4 --tile 4096 0,0,0,0
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --graph osc --tile 8192 0,0,0,0
done
done
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
