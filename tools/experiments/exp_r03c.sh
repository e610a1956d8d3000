#!/bin/bash
# round 3, third exploration: time-major lockstep refinement (rows in flight, prefetch distance, XCD map), other stream counts and a
# 4-wire graph; the launch bubble of config 2 (block length scan)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03c; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1
L=524288
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 4,1,1024,$L 4,1,1024,$((L+32)) 2,1,1024,$L 2,2,1024,$L 2,4,1024,$L 2,2,1024,$((L+32)) 4,1,512,$L 4,2,512,$L 4,1,1024,$((L+2)) 4,1,1024,$((L+1)) 2,4,512,$L 1,4,1024,$L 1,8,1024,$L > $O/timemajor_1M.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 262144 --tile 0 --rounds 15 0,0 2,16 4,1,1024,$L 2,2,1024,$L 2,4,1024,$L 4,4,512,$L 1,8,1024,$L > $O/timemajor_256k.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 524288 --tile 0 --rounds 9 0,0 2,16 4,1,1024,$L 2,2,1024,$L 2,4,1024,$L 4,4,512,$L > $O/timemajor_512k.txt 2>&1
python tools/sweep.py --graph par4 --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 1,16 1,2,1024,$L 1,4,1024,$L 1,8,1024,$L 1,4,512,$L 2,2,1024,$L > $O/timemajor_par4.txt 2>&1
python tools/sweep.py --graph osc --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 2,16 4,1,1024,$L 2,2,1024,$L 2,4,1024,$L 2,4,512,$L > $O/timemajor_osc.txt 2>&1
for T in 2048 4096 8192 16384; do
python tools/sweep.py --graph cascade6 --streams 65536 --samples $T --tile 8192 --rounds 30 0,0 1,16,256,34816 1,16,256,2048 1,16,0,32768 > $O/config2_T$T.txt 2>&1
done
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,24,256,8 1,32,256,8 1,16,256,34816 1,16,256,2048 1,16,0,32768 > $O/config2.txt 2>&1
echo done
