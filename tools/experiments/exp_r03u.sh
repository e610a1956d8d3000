#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
for opts in "" "-DFZ_DBG_NOLOAD" "-DFZ_DBG_NOSTORE" "-DFZ_DBG_NOLOAD -DFZ_DBG_NOSTORE"; do
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 300 python tools/experiments/exp_r03u.py 2>&1 | grep -v amdgpu
done > $O/out.txt 2>&1
cat $O/out.txt
