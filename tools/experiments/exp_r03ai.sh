#!/bin/bash
# round 3: v_pk_mov_b32 pair assembly in the wave-split rounds -- parity of the wave-split tests, then the few-stream sweeps
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03ai; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave or stage_pack or few_streams" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 2>&1 | grep -v amdgpu.ids > $O/config2.txt
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,1024 1,16,128,34816 1,32,128,34816 2>&1 | grep -v amdgpu.ids > $O/config2h.txt
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,16,64,34816 1,32,64,2048 2>&1 | grep -v amdgpu.ids > $O/config2q.txt
tail -n +1 $O/config2*.txt
