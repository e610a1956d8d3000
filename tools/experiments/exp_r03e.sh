#!/bin/bash
# round 3: sub-atom stage packing (parity: the whole GPU suite) + few-stream sweeps + translation touch-ahead on time-major frames
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03e; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 1,24,256,8 > $O/config2.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 0,0 1,32,128,1024 1,16,128,34816 1,32,128,34816 1,16,128,2048 1,32,128,2048 1,16,256,8 > $O/config2h.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 0,0 1,16,64,34816 1,32,64,2048 1,16,64,2048 1,32,128,33792 > $O/config2q.txt 2>&1
L=524288
T1=$((1<<24)); T2=$((2<<24)); T3=$((3<<24)); T4=$((4<<24)); T5=$((5<<24))
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 2,2,1024,$((L+T1)) 2,2,1024,$((L+T2)) 2,2,1024,$((L+T3)) 2,2,1024,$((L+T4)) 2,2,1024,$((L+T5)) 2,16,256,$T3 2,16,256,0 2,4,1024,$((L+T3)) 1,8,1024,$((L+T3)) 2,8,1024,$((L+T3)) 2,8,512,$((L+T3)) 2,16,256,$T5 > $O/timemajor_touch.txt 2>&1
python tools/stream_major_bench.py > $O/stream_major.txt 2>&1
echo done
