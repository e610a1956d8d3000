#!/bin/bash
# round 3, second exploration: wave splits with the tuple-major / rotated role map; time-major frames in lockstep (block x P x U)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03b; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 30 0,0 1,16,256,34816 1,8,256,34816 1,16,128,34816 1,16,64,34816 1,16,256,33792 1,16,128,33792 1,16,0,32768 1,16,256,2048 > $O/config2.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 30 0,0 1,32,128,1024 1,16,128,34816 1,16,256,34816 1,16,256,33792 1,32,64,34816 > $O/config2h.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 30 0,0 1,16,64,34816 1,32,64,34816 1,16,128,34816 > $O/config2q.txt 2>&1
L=524288
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 2,16 4,1,1024,$L 2,4,1024,$L 2,8,1024,$L 1,16,1024,$L 4,4,768,$L 4,8,768,$L 2,8,768,$L 2,16,768,$L 4,4,512,$L 4,8,512,$L 4,16,512,$L 2,8,1024,0 4,1,1024,0 4,8,512,0 > $O/timemajor.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 8192 --rounds 7 0,0 2,16 2,16,256,2097152 4,1,1024,$L 2,8,1024,$L 4,8,512,$L 2,8,1024,0 > $O/tiled.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 0 --rounds 20 0,0 1,16,256,8 1,16,256,34816 1,16,0,32768 > $O/config2_timemajor.txt 2>&1
echo done
