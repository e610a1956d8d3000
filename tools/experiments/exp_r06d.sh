#!/usr/bin/env bash
# Round 6, batch d:   gpurun --timeout 1500 -- 'bash tools/experiments/exp_r06d.sh'   -> gpurun_out/r06d/
#  LDS rings in lockstep on six fresh allocations under other cache policies of the frame loads / stores (cpol: 0 plain, 2 nt, 16 sc1, 18 nt|sc1 = default store, 17 sc0|sc1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1 FLOWZ_HIP_CACHE=/tmp/fz_kc_r06d
for opts in "" "-DFZ_DBG_AUX_ST=0" "-DFZ_DBG_AUX_ST=2" "-DFZ_DBG_AUX_ST=16" "-DFZ_DBG_AUX_LD=0" "-DFZ_DBG_AUX_LD=0 -DFZ_DBG_AUX_ST=0" "-DFZ_DBG_AUX_LD=1"; do
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 600 python tools/experiments/exp_r06d.py >> $O/ldsring_policies.txt 2>&1
done
grep -v amdgpu.ids $O/ldsring_policies.txt | cut -c1-400
# randomized parity runs on the final kernels (tools/fuzz_*.py): random graphs with comparison operators among them, whole waves (lane groups), wide frames, wave splits
export FLOWZ_HIP_CACHE=/tmp/fz_kc_fuzz
FUZZ_CMP=1 timeout 420 python tools/fuzz_gpu.py 60000 400 300 > $O/fuzz_gpu_cmp_ns200.txt 2>&1; tail -2 $O/fuzz_gpu_cmp_ns200.txt
FUZZ_CMP=1 FUZZ_NS=512 timeout 420 python tools/fuzz_gpu.py 61000 400 300 > $O/fuzz_gpu_cmp_ns512.txt 2>&1; tail -2 $O/fuzz_gpu_cmp_ns512.txt
timeout 300 python tools/fuzz_gpu.py 62000 300 200 > $O/fuzz_gpu_ns200.txt 2>&1; tail -2 $O/fuzz_gpu_ns200.txt
timeout 300 python tools/fuzz_wide_frames.py 63000 300 200 > $O/fuzz_wide_frames.txt 2>&1; tail -2 $O/fuzz_wide_frames.txt
timeout 300 python tools/fuzz_wave_split.py 64000 100 200 > $O/fuzz_wave_split.txt 2>&1; tail -2 $O/fuzz_wave_split.txt
