#!/bin/bash
# (NOTE: the kernel / generator knob this script drives was an experiment and has been taken out again -- profiles/NOTES.md, "What the lone waves wait for"; kept as the record of what was run)
# GPU box, round 4: the pair body with the waves of a CU started a quarter of a phase apart (-DFZ_DBG_STAGGER): 262 144 streams are two rounds of waves that
# all start together, so the in-runs and out-runs of the whole chip come in bursts.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04aa; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
for opt in "" "-DFZ_DBG_STAGGER=0" "-DFZ_DBG_STAGGER=1"; do
echo "## EXTRA_OPTS=[$opt]"
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --sm --streams 262144 0,0,0,0
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --sm --streams 131072 0,0,0,0
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --sm 0,0,0,0
done
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
