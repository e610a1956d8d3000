#!/bin/bash
# round 3, first exploration: (A) config 2 with three parts + I/O wave at 4 tuples per CU, (B) time-major frames in lockstep, (C) stream-major baseline
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03a; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 30 0,0 1,16,256,8 1,16,64,34816 1,8,64,34816 1,16,64,2048 1,32,64,2048 1,16,0,32768 1,16,128,33792 > $O/config2.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 30 0,0 1,16,128,1024 1,32,128,1024 1,16,64,34816 1,16,64,2048 1,32,64,2048 1,32,64,34816 > $O/config2h.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 30 0,0 1,16,64,34816 1,32,64,34816 1,16,64,2048 1,32,64,2048 > $O/config2q.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 7 0,0 2,16 2,32 4,8 2,16,256,524288 2,16,512,524288 2,8,1024,524288 4,8,256,524288 4,8,512,524288 4,4,1024,524288 2,32,512,524288 2,8,512,524288 > $O/timemajor.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 8192 --rounds 7 0,0 2,16 2,16,256,2097152 2,16,512,524288 2,16,256,524288 4,8,256,1048576 > $O/tiled.txt 2>&1
python tools/stream_major_bench.py > $O/stream_major.txt 2>&1
echo done
