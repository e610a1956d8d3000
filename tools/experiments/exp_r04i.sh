#!/bin/bash
# GPU box, round 4: (1) LDS ring reads of a chunk fetched together (ring_prefetch) against reads in place (-DFZ_DBG_NO_RING_PREFETCH is
# not needed: chunks of 8 rows and rings read 9..15 samples back still read in place) -- the lds_ring graph of the bench line;
# (2) typed frames with the plain store policy as the default; (3) the oscillator chain's new time-major default.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04i; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896; LGP=8912928
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph ldsring 0,0,0,0 2,16,128,0 1,16,128,0 1,16,256,0 2,8,128,0 2,32,128,0 1,32,256,0 1,16,64,0 2,16,64,0
$S --graph c32onepole 0,0,0,0 4,1,1024,$LGP 2,2,1024,$LG 2,16,256,0
$S --graph f64biquad 0,0,0,0 4,1,1024,$LGP 2,2,1024,$LG 2,16,256,0
$S --graph osc 0,0,0,0 2,1,1024,$LGP
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "lds or ring or delay or typed or complex or double or f64 or graphs_vs_oracle or fuzz or random" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.txt
tail -5 $O/pytest_sel.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
