#!/usr/bin/env bash
# Round 6, batch m:   gpurun --timeout 900 -- 'bash tools/experiments/exp_r06m.sh'   -> gpurun_out/r06m/
#  stream-major buffers, 1 M streams x 4096 (pair long-run body) and 65 536 streams (one-stream long-run body): other cache policies of the frame accesses
#  (cpol: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0|sc1, 18 nt|sc1 = the default store; default load nt = 2), one process per policy, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_CACHE=/tmp/fz_kc_r06m
for pass in 1 2; do
for opts in "" "-DFZ_DBG_AUX_ST=2" "-DFZ_DBG_AUX_ST=16" "-DFZ_DBG_AUX_ST=0" "-DFZ_DBG_AUX_LD=0" "-DFZ_DBG_AUX_LD=1" "-DFZ_DBG_AUX_LD=0 -DFZ_DBG_AUX_ST=2"; do
  echo "## extra opts: '$opts' (pass $pass)" >> $O/stream_major_policies.txt
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 300 python tools/sweep.py --graph cascade6 --sm --rounds 5 --reps 3 0,0 >> $O/stream_major_policies.txt 2>&1
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 300 python tools/sweep.py --graph cascade6 --sm --streams 65536 --rounds 5 --reps 40 0,0 >> $O/stream_major_policies.txt 2>&1
done; done
grep -v amdgpu.ids $O/stream_major_policies.txt | grep "^##\|^{" | cut -c1-150
