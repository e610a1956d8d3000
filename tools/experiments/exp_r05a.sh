#!/bin/bash
# Round 5, first sweeps (GPU box; `PREBUILD=1 bash tools/experiments/exp_r05a.sh` here first: builds the kernels without a GPU).
#  1. config 2 (65 536 x 4096) on plain time-major rows and tiles: workgroup sizes, lockstep without the XCD-wide step, I/O waves, store policies
#  2. typed frames: the 4-bytes-in / 8-bytes-out yardsticks (one operation) and the two bench graphs: lane packings x lockstep x store policies
#  3. LDS rings vectorised in time: packings x chunk lengths
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-exp_r05a}; mkdir -p $O
PB=${PREBUILD:+--prebuild}
LS=524288; GS=8388608; P3=32; SP=8; IO=32768; IO2=33554432
run() { python tools/sweep.py $PB "$@" 2>&1 | grep -v "^$" ; }
{
echo "## config 2 time-major"
run --graph cascade6 --streams 65536 --reps 20 --rounds 5 0,0 1,16,256,$SP 1,16,128,$SP 1,16,64,$SP 1,16,512,$SP 1,8,256,$SP 1,32,256,$SP 1,8,64,$SP 1,32,64,$SP \
    1,16,256,$((SP+LS)) 1,16,0,$IO 1,16,0,$((IO+IO2)) 1,16,256,$((SP+393216)) 1,16,256,$((SP+458752)) 1,16,64,$((SP+393216)) 1,16,64,$((SP+458752))
echo "## config 2 tiled 8192"
run --graph cascade6 --streams 65536 --tile 8192 --reps 20 --rounds 5 0,0 1,16,256,$SP 1,16,128,$SP 1,16,64,$SP 1,16,0,$IO 1,16,0,$((IO+IO2)) 1,16,64,$((SP+393216))
for g in widen64 widenc32 f64biquad c32onepole; do
  echo "## typed $g"
  run --graph $g --streams 1048576 --reps 3 --rounds 3 0,0 2,16,256 4,8,256 4,4,256 2,2,1024,$((LS+GS)) 2,1,1024,$((LS+GS+P3)) 4,1,1024,$((LS+GS+P3)) 4,2,512,$((LS+GS)) \
      2,16,256,393216 2,16,256,458752 2,16,256,196608 2,2,1024,$((LS+GS+393216)) 2,2,1024,$((LS+GS+458752)) 4,1,1024,$((LS+GS+P3+393216)) 4,1,1024,$((LS+GS+P3+458752))
done
echo "## LDS rings"
run --graph ldsring --streams 1048576 --reps 3 --rounds 3 0,0 1,32,256 2,32,128 1,16,256 2,16,128 1,32,128 1,16,128 2,32,64 2,16,64 1,32,512 4,16,64 4,16,128
} > $O/sweeps.txt 2>&1
cat $O/sweeps.txt | cut -c1-220
