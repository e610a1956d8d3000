#!/bin/bash
# round 3: GPU suite after the line-depth fix; max-ilp scheduling for the low-ILP wave-split kernels
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03f; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
for mode in base ilp; do
  if [ $mode = ilp ]; then export FLOWZ_HIP_EXTRA_OPTS="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
  python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 1,24,256,8 > $O/config2_$mode.txt 2>&1
  python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,1024 1,16,128,34816 1,16,128,2048 1,16,128,33792 > $O/config2h_$mode.txt 2>&1
  python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,16,64,34816 1,32,64,2048 1,16,64,2048 > $O/config2q_$mode.txt 2>&1
  python tools/sweep.py --graph osc --streams 32768 --tile 8192 --rounds 40 0,0 1,16,256,8 > $O/osc32k_$mode.txt 2>&1
done
L=524288
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 2,2,1024,$((L+(2<<24))) 4,1,1024,$((L+32)) 4,1,1024,$((L+32+(2<<24))) 4,1,1024,$((L+32+(3<<24))) 2,16,256,0 > $O/timemajor.txt 2>&1
echo done
