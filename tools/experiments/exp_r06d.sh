#!/usr/bin/env bash
# Round 6, batch d:   gpurun --timeout 1500 -- 'bash tools/experiments/exp_r06d.sh'   -> gpurun_out/r06d/
#  LDS rings in lockstep on six fresh allocations under other cache policies of the frame loads / stores (cpol: 0 plain, 2 nt, 16 sc1, 18 nt|sc1 = default store, 17 sc0|sc1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_CACHE=/tmp/fz_kc_r06d
for opts in "" "-DFZ_DBG_AUX_ST=0" "-DFZ_DBG_AUX_ST=2" "-DFZ_DBG_AUX_ST=16" "-DFZ_DBG_AUX_LD=0" "-DFZ_DBG_AUX_LD=0 -DFZ_DBG_AUX_ST=0" "-DFZ_DBG_AUX_LD=1"; do
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 600 python tools/experiments/exp_r06d.py >> $O/ldsring_policies.txt 2>&1
done
grep -v amdgpu.ids $O/ldsring_policies.txt | cut -c1-400
