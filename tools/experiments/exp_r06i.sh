#!/usr/bin/env bash
# Round 6, batch i (verification after 'tiles in lockstep'):   gpurun --timeout 3000 -- "FZ_COMMIT=<sha> bash tools/experiments/exp_r06i.sh"
#  the whole GPU suite + the default bench line (tools/gpu_round.sh), then the PMC passes of the bench objects whose default kernel changed since prof_r06 (tiled cascade, lds_ring)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_round.sh r06i
PASSES_ONLY=1 ONLY_TAGS='lds_ring|cascade6_1048576-tiled' bash tools/profile_bench.sh prof_r06i
