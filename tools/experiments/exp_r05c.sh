#!/bin/bash
# Round 5, second sweeps (PREBUILD=1 here first).  LDS rings with more rows in flight; wide frames / register-heavy graphs on plain time-major rows.
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-exp_r05c}; mkdir -p $O
PB=${PREBUILD:+--prebuild}
LS=524288; GS=8388608; P3=32; SP=8
run() { python tools/sweep.py $PB "$@" 2>&1 | grep -v "^$\|amdgpu.ids" ; }
{
echo "## LDS rings: rows in flight"
run --graph ldsring --streams 1048576 --reps 3 --rounds 3 1,32,256 1,32,256,$P3 1,16,256,$P3 1,64,256 1,64,128 2,16,128,$P3 1,8,256,$P3
echo "## 4-wire sum, time-major"
run --graph par4 --streams 1048576 --reps 2 --rounds 3 0,0 1,2,1024,$((LS+GS)) 1,1,1024,$((LS+GS+P3)) 2,1,1024,$((LS+GS+P3)) 2,2,512,$((LS+GS)) 2,1,512,$((LS+GS+P3)) 1,32,256 1,16,256 1,4,512,$((LS+GS)) 1,2,512,$((LS+GS))
echo "## oscillator chain, time-major"
run --graph osc --streams 1048576 --reps 2 --rounds 3 0,0 1,4,1024,$((LS+GS+SP)) 2,1,1024,$((LS+GS+P3)) 2,2,512,$((LS+GS)) 2,16,256 1,16,256,$SP 2,8,256
[ -z "$PB" ] && { echo "## allocation placement"; python tools/experiments/exp_r05b.py; }
} > $O/sweeps.txt 2>&1
cat $O/sweeps.txt | cut -c1-220
