#!/bin/bash
# GPU box, round 4: (1) stream-major buffers, wide frames in / narrow frames out (the 4-wire sum of config 3): outputs of n_in / n_out chunks held
# in registers and stored as out-runs as long as the in-runs (FZ_SM_HOLD) against one 128-byte out-run per chunk (-DFZ_DBG_NO_HOLD);
# (2) which grid the output rows must start on for the write-through store policy (FZ_VF_ST_MERGE: 1 000 004 / 1 000 008 / 1 000 016 /
# 1 000 032 streams = rows on the 16 / 32 / 64 / 128-byte grid), and nt-only stores on rows ON the grid; (3) the new defaults of r04k.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04l; mkdir -p $O
MODE=${1:-run}
LGP=8912928; NT=$((7<<16)); SC=$((6<<16))
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --sm --graph par4 0,0,0,0 1,16,256,0 1,16,128,0 1,8,256,0
FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_NO_HOLD $S --sm --graph par4 0,0,0,0 1,16,256,0
$S --sm --graph par4 --streams 65536 0,0,0,0 1,16,256,0
FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_NO_HOLD $S --sm --graph par4 --streams 65536 0,0,0,0
for n in 1000004 1000008 1000016 1000032; do $S --streams $n 4,1,1024,$LGP 4,1,1024,$((LGP+NT)) 4,1,1024,$((LGP+SC)); done
$S --streams 1048576 4,1,1024,$LGP 4,1,1024,$((LGP+NT))
$S --streams 1048577 0,0,0,0
$S --graph ldsring 0,0,0,0
$S --sm --streams 65536 0,0,0,0
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "stream_major or ragged" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.txt
tail -5 $O/pytest_sel.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
