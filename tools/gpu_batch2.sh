#!/usr/bin/env bash
# Dev tool (GPU box): stream-major access-pattern microbenchmark + prefetch-depth sweep of config 2.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-batch2}; mkdir -p $O; cd $R
tools/_bin/sm_bench > $O/sm_bench.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 1,32,256,8 1,8,256,40 1,12,256,40 1,16,256,40 1,24,256,40 1,32,256,40 > $O/sweep_config2_prefetch.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 1,12,256,40 1,16,256,40 1,24,256,40 > $O/sweep_32k_prefetch.txt 2>&1
cat $O/sm_bench.txt $O/sweep_config2_prefetch.txt $O/sweep_32k_prefetch.txt
