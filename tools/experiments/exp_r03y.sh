#!/usr/bin/env bash
# round 3: compiler scheduling strategies for the long-run stream-major kernel (as shipped and without memory traffic)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
for s in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-mllvm -amdgpu-use-amdgpu-trackers=1" "-mllvm -amdgpu-schedule-metric-bias=0"; do
  for m in "" "-DFZ_DBG_NOLOAD -DFZ_DBG_NOSTORE"; do
    FLOWZ_HIP_EXTRA_OPTS="$s $m" timeout 200 python tools/experiments/exp_r03u.py 2>&1 | grep "^cascade6"
  done
done > $O/out.txt 2>&1
cat $O/out.txt
