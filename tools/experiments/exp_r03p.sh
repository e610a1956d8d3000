#!/bin/bash
# round 3: XCD-synchronised lockstep workgroups on stream tiles of various sizes (is a tile of one XCD's worth of streams better than plain rows?)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03p; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
G=8388608; L=524288; V4=$((L+32+G)); V2=$((L+G))
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 9 0,0 2,2,1024,$V2 > $O/tile0.txt 2>&1
for t in 8192 32768 65536 131072 262144 524288; do
  python tools/sweep.py --graph cascade6 --streams 1048576 --tile $t --rounds 9 0,0 4,1,1024,$V4 2,2,1024,$V2 > $O/tile$t.txt 2>&1
done
python tools/sweep.py --graph par4 --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 1,2,1024,$V2 1,4,1024,$V2 1,1,1024,$((V2+32)) > $O/par4_tm.txt 2>&1
python tools/sweep.py --graph par4 --streams 1048576 --samples 2048 --tile 4096 --rounds 5 0,0 1,2,1024,$V2 > $O/par4_tiled.txt 2>&1
python tools/sweep.py --graph osc --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 1,4,1024,$V2 2,16,256,0 > $O/osc_tm.txt 2>&1
grep -hv amdgpu $O/*.txt | cut -c1-200
