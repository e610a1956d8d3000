#!/bin/bash
# GPU box, round 4 first look: the whole new bench line, then variant sweeps for the shapes the line adds
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a; mkdir -p $O
( time timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.txt ) 2> $O/bench_time.txt; echo "bench rc=$?"
tail -c 1500 $O/bench_err.txt; cat $O/bench_time.txt
python tools/show_bench.py $O/bench_line.json > $O/bench_summary.txt 2>&1; cat $O/bench_summary.txt
export FLOWZ_HIP_ISOLATED_HIPRTC=1 FLOWZ_HIP_AUTOTUNE=0
S="timeout 600 python tools/sweep.py --rounds 7"
{
$S --streams 786432  4,1,1024,8912928 4,1,768,8912928 2,2,1024,8912896 2,16,256,0 1,4,1024,8912896 4,8,256,1048576
$S --streams 2097152 4,1,1024,8912928 2,2,1024,8912896 2,16,256,0 4,8,256,1048576 2,16,256,2097152
$S --streams 1048577 1,16,256,8 1,4,1024,8912904 1,4,1024,8912896 1,16,256,16 1,8,1024,524296
$S --streams 1000000 4,1,1024,8912928 2,2,1024,8912896 2,16,256,0
$S --graph osc 1,4,1024,8912896 2,8,512,524288 2,2,512,8912896 2,16,256,0 1,16,256,0 1,8,1024,524288 tune
$S --graph par4 1,4,1024,8912896 1,2,1024,8912896 1,32,256,0 1,16,256,0 1,8,1024,524288 tune
$S --streams 65536 1,16,256,8 1,16,0,32768 1,24,256,8 1,16,256,3072 tune
} > $O/sweeps.txt 2>&1
cat $O/sweeps.txt
