#!/bin/bash
# round 3: per-XCD synchronisation of the lockstep workgroups (FZ_VF_GRID_SYNC = 8388608): time-major and tiled frames, stream counts
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03n; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_AUTOTUNE=0
python - > $O/parity.txt 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from zignal_amd import flowz as F, workloads as W
p=F.compile(F.from_sexpr(W.df1_cascade(6)))
for ns,T in ((1<<20,300),((1<<21)+4096*3,64),(1<<18,200)):
    x=torch.empty((T,ns,1),device='cuda'); F.synth_fill(x,1)
    y0,s0=p.run_block(x,variant=F.make_variant(2,16,256,0))
    for v in ((4,1,1024,524320+8388608),(2,2,1024,524288+8388608),(1,8,1024,524288+8388608)):
        y,s=p.run_block(x,variant=F.make_variant(*v)); print(ns,T,v, torch.equal(y,y0), torch.equal(s,s0))
PY
cat $O/parity.txt
G=8388608; L=524288
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 0 --rounds 9 0,0 4,1,1024,$((L+32+G)) 4,1,1024,$((L+G)) 2,2,1024,$((L+G)) 1,8,1024,$((L+G)) 2,4,1024,$((L+G)) 2,16,256,0 > $O/timemajor.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 1048576 --tile 8192 --rounds 9 0,0 4,1,1024,$((L+32+G)) 2,2,1024,$((L+G)) 4,1,1024,$((L+32)) > $O/tiled.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 2097152 --tile 0 --rounds 5 0,0 4,1,1024,$((L+32+G)) > $O/timemajor_2M.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 524288 --tile 0 --rounds 9 0,0 2,2,1024,$((L+G)) 1,8,1024,$((L+G)) > $O/timemajor_512k.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 262144 --tile 0 --rounds 9 0,0 1,8,1024,$((L+G)) 1,4,1024,$((L+G)) > $O/timemajor_256k.txt 2>&1
python tools/sweep.py --graph osc --streams 1048576 --tile 0 --samples 2048 --rounds 5 0,0 2,16,256,0 1,8,1024,$((L+G)) > $O/timemajor_osc.txt 2>&1
python tools/sweep.py --graph par4f --streams 1048576 --tile 0 --samples 2048 --rounds 5 0,0 4,1,1024,$((L+32+G)) > $O/timemajor_par4f.txt 2>&1
grep -hv amdgpu $O/timemajor*.txt $O/tiled.txt | cut -c1-200
