#!/usr/bin/env bash
# Round 6, batch f (verification of the final tree):   gpurun --timeout 3000 -- "FZ_COMMIT=<sha> bash tools/experiments/exp_r06f.sh"   -> gpurun_out/r06f/, gpurun_out/prof_r06f/
#  the whole GPU suite + the default bench line (tools/gpu_round.sh), then the PMC passes of the one bench object whose default kernel changed after prof_r06 (lds_ring: three chunk buffers)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_round.sh r06f
PASSES_ONLY=1 ONLY_TAGS='lds_ring' bash tools/profile_bench.sh prof_r06f
