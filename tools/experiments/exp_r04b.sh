#!/bin/bash
# GPU box, round 4: (A) buffer range checks / misaligned rows (tools/oob_probe.hip); (B) time-major launch geometry -- streams per lane x
# lanes per workgroup x laps -- over stream counts; (C) register-heavy graphs stage-packed in lockstep; (D) config 2 on time-major rows;
# (E) an odd stream count under the store cache policies; (F) the modulated cascade, LDS rings, 64-sample windows.
# usage: exp_r04b.sh [prebuild]   (prebuild: on the GPU-less box, compiles every kernel the sweeps launch into the cache)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04b; mkdir -p $O
MODE=${1:-run}
export FLOWZ_HIP_AUTOTUNE=0
L=524288; LG=8912896; LGP=8912928; LGS=8912904; LS=524296
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
sweeps() {
$S --streams 262144  1,4,1024,$LG 2,4,512,$LG 2,2,512,$LG 4,4,256,$LG 4,2,256,$LG 2,16,256,0
$S --streams 393216  4,2,384,$LG 4,4,384,$LG 2,2,768,$LG 1,4,1024,$LG 2,16,256,0
$S --streams 524288  2,2,1024,$LG 4,2,512,$LG 4,1,512,$LGP 4,4,512,$LG
$S --streams 786432  4,1,768,$LGP 4,2,768,$LG
$S --streams 1000000 4,1,1024,$LGP 4,1,960,$LGP
$S --streams 1048576 4,1,1024,$LGP
$S --streams 1572864 4,1,768,$LGP 4,1,1024,$LGP 2,2,1024,$LG
$S --streams 2097152 4,1,1024,$LGP 2,2,1024,$LG
$S --graph osc 1,8,1024,$L 1,4,1024,$LG 1,4,1024,$LGS 1,8,1024,$LS 1,2,1024,$LGS 1,16,256,8
$S --streams 65536 1,16,256,8 1,16,256,10 1,8,256,8 1,32,256,8 1,16,256,40 1,16,256,34816 1,16,256,2048 1,16,256,65544 1,16,256,458760
$S --streams 1048577 1,16,256,8 1,16,256,65544 1,16,256,458760 1,16,256,196616 1,4,1024,8978440 1,4,1024,9371656 1,16,256,65552
$S --graph mod6 4,1,1024,$LGP 2,2,1024,$LG 2,16,256,0 4,8,256,1048576
$S --graph ldsring 2,16,128,0 1,16,256,0 2,8,128,0 1,8,256,0 2,16,64,0 1,16,128,0 1,8,128,0
$S --graph params6 --samples 64 1,4,1024,$LG 2,16,256,0 1,16,256,0 2,8,256,0 4,4,256,0 1,4,1024,$L
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
tools/_bin/oob_probe > $O/oob_probe.txt 2>&1; cat $O/oob_probe.txt
sweeps > $O/sweeps.txt 2>&1
( export FLOWZ_HIP_LAPS=kernel
$S --streams 2097152 4,1,1024,$LGP 2,2,1024,$LG
$S --graph osc 1,4,1024,$LG 1,4,1024,$LGS ) > $O/sweeps_kernel_laps.txt 2>&1
cat $O/sweeps.txt $O/sweeps_kernel_laps.txt
