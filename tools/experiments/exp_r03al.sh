#!/bin/bash
# round 3: the step's operations list-scheduled by the generator and pinned (FLOWZ_HIP_PIN_ORDER=1) against the compiler's order
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03al; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
for pin in 0 1; do
  export FLOWZ_HIP_PIN_ORDER=$pin
  python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,32,64,2048 1,16,64,34816 2>&1 | grep -v amdgpu.ids > $O/config2q_pin$pin.txt
  python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,1024 1,32,128,34816 2>&1 | grep -v amdgpu.ids > $O/config2h_pin$pin.txt
  python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 2>&1 | grep -v amdgpu.ids > $O/config2_pin$pin.txt
done
FLOWZ_HIP_PIN_ORDER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave or stage_pack or few_streams" > $O/pytest_pin1.txt 2>&1; tail -3 $O/pytest_pin1.txt
tail -n +1 $O/config2*.txt
