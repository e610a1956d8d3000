#!/bin/bash
# (NOTE: the kernel / generator knob this script drives was an experiment and has been taken out again -- profiles/NOTES.md, "What the lone waves wait for"; kept as the record of what was run)
# GPU box, round 4: LDS ring reads shorter than the chunk forwarded from registers (no step of a chunk waits for an LDS round trip): parity of every
# test that touches rings / delays, then the lds_ring graph of the bench line.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04ab; mkdir -p $O
MODE=${1:-run}
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph ldsring 0,0,0,0 1,32,256,0 2,16,128,0 2,32,64,0 1,32,128,0 1,16,256,0 4,16,64,0
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_graphs.py -q -x -k "lds or ring or delay or graphs_vs_oracle or random or far" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.txt
tail -4 $O/pytest_sel.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
