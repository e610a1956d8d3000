#!/bin/bash
# (NOTE: the kernel / generator knob this script drives was an experiment and has been taken out again -- profiles/NOTES.md, "What the lone waves wait for"; kept as the record of what was run)
# GPU box, round 4: what the lone waves of the few-stream frame kernels wait for.  (1) -DFZ_DBG_PRIME_VMCNT: dummy stores in the preheader so that the waitcnt
# pass's merged state at the loop header is exact; (2) -DFZ_DBG_KEEP_STORE_REGS: output registers of their own for every row of the chunk buffers
# (a frame store's data registers are guarded by vmcnt: re-used registers wait for the store to COMPLETE); (3) store policies with fast acknowledgement.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04ad; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 11"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
NT=$((7<<16)); PL=$((1<<16))
sweeps() {
for opt in "" "-DFZ_DBG_PRIME_VMCNT" "-DFZ_DBG_KEEP_STORE_REGS" "-DFZ_DBG_PRIME_VMCNT -DFZ_DBG_KEEP_STORE_REGS"; do
echo "## EXTRA_OPTS=[$opt]"
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --streams 65536 --tile 8192 1,16,256,8 1,16,256,$((8+NT)) 1,16,256,$((8+PL)) 1,32,256,8
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --streams 65536 1,16,256,8
FLOWZ_HIP_EXTRA_OPTS="$opt" $S --graph ldsring 0,0,0,0
done
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
