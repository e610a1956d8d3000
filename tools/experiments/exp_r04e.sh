#!/bin/bash
# GPU box, round 4: the cross-paired wave split (FZ_VF_CROSS_PAIR): parity tests, then few-stream sweeps against today's defaults;
# LDS rings exact / power of two; the remainder next to a lap with the lap's waves at high priority.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04e; mkdir -p $O
MODE=${1:-run}
X=16812032
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
sweeps() {
export FLOWZ_HIP_AUTOTUNE=0
$S --streams 16384 --tile 8192 0,0,0,0 1,16,64,$X 1,32,64,$X 1,8,64,$X 1,16,64,34816
$S --streams 32768 --tile 8192 0,0,0,0 1,16,128,$X 1,16,64,$X 1,8,128,$X 1,32,64,$X
$S --streams 65536 --tile 8192 0,0,0,0 1,8,256,$X 1,8,128,$X 1,16,128,$X
$S --streams 8192 --tile 8192 0,0,0,0 1,16,64,$X
$S --streams 16384 0,0,0,0 1,16,64,$X
$S --streams 65536 0,0,0,0 1,8,256,$X
$S --graph ldsring 0,0,0,0 2,16,256,0 2,16,128,0 1,16,128,0 1,16,256,0
$S --streams 1048577 0,0,0,0
$S --streams 1048576 0,0,0,0
}
if [ "$MODE" = prebuild ]; then sweeps; FLOWZ_HIP_LDS_POW2=1 $S --graph ldsring 0,0,0,0 2,16,256,0 2,16,128,0 1,16,128,0; exit 0; fi
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "cross" > $O/pytest_cross.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_cross.txt
tail -15 $O/pytest_cross.txt
sweeps > $O/sweeps.txt 2>&1
( export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_LDS_POW2=1; $S --graph ldsring 0,0,0,0 2,16,256,0 2,16,128,0 1,16,128,0 ) > $O/sweeps_pow2.txt 2>&1
( export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_EXTRA_OPTS="-mllvm -amdgpu-sched-strategy=max-ilp"; $S --streams 16384 --tile 8192 1,16,64,$X 1,32,64,$X; $S --streams 32768 --tile 8192 1,16,128,$X ) > $O/sweeps_maxilp.txt 2>&1
( export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_EXTRA_OPTS="-mllvm -amdgpu-sched-strategy=iterative-ilp"; $S --streams 16384 --tile 8192 1,16,64,$X; $S --streams 32768 --tile 8192 1,16,128,$X ) > $O/sweeps_iterilp.txt 2>&1
grep -hv amdgpu.ids $O/sweeps.txt $O/sweeps_pow2.txt $O/sweeps_maxilp.txt $O/sweeps_iterilp.txt
