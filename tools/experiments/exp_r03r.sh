#!/bin/bash
# round 3: persistent launch + XCD-wide synchronisation beyond one lap and on the other HBM-bound kernels
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03r; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
python - > $O/parity.txt 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from zignal_amd import flowz as F, workloads as W
G=8388608; L=524288
p=F.compile(F.from_sexpr(W.df1_cascade(6)))
for ns,T in (((1<<21)+4096*3+8,70),(300*1024+64,130),(65536,300),(65536+512,77)):
    x=torch.empty((T,ns,1),device='cuda'); F.synth_fill(x,1)
    y0,s0=p.run_block(x,variant=F.make_variant(1,8,256,16))
    for v in ((4,1,1024,L+32+G),(2,2,1024,L+G),(1,4,1024,L+G),(1,16,256,8+L+G),(1,8,256,8+L+G),(2,16,256,L+G),(1,8,64,L+G)):
        if ns % v[0]: continue
        y,s=p.run_block(x,variant=F.make_variant(*v)); print(ns,T,v, torch.equal(y,y0), torch.equal(s,s0))
PY
cat $O/parity.txt
G=8388608; L=524288
python tools/sweep.py --graph cascade6 --streams 2097152 --tile 0 --rounds 5 0,0 4,1,1024,$((L+32)) > $O/tm_2M.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 0,0 1,16,256,$((8+L+G)) 1,16,256,$((8+L)) 1,24,256,$((8+L+G)) > $O/config2_tiled.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 0 --rounds 40 0,0 1,16,256,$((8+L+G)) 1,16,256,$((8+L)) > $O/config2_tm.txt 2>&1
python tools/sweep.py --graph par4 --streams 1048576 --samples 2048 --tile 4096 --rounds 5 0,0 1,4,1024,$((L+G)) 1,2,1024,$((L+G)) 1,16,256,$((L+G)) 1,32,256,$((L+G)) > $O/par4_tiled.txt 2>&1
python tools/sweep.py --graph par4 --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 1,4,1024,$((L+G)) 1,2,1024,$((L+G)) 1,1,1024,$((L+G+32)) > $O/par4_tm.txt 2>&1
python tools/sweep.py --graph osc --streams 1048576 --samples 2048 --tile 8192 --rounds 5 0,0 2,16,256,$((L+G)) 1,4,1024,$((L+G)) > $O/osc_tiled.txt 2>&1
python tools/sweep.py --graph osc --streams 1048576 --samples 2048 --tile 0 --rounds 5 0,0 2,16,256,$((L+G)) 2,16,256,0 > $O/osc_tm.txt 2>&1
grep -hv amdgpu $O/tm_2M.txt $O/config2*.txt $O/par4*.txt $O/osc*.txt | cut -c1-200
